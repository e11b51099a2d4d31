"""The reference's stdout grammar (src/find_mems.cpp:115-118,138 + src/tag_arrays.cpp:885-889) applied to
an oracle batch result -- test infrastructure for the CLI parity test."""


def format_find_mems(res):
    out = []
    mo, po = res["mem_offsets"], res["pos_offsets"]
    for i in range(len(mo) - 1):
        out.append("Seq: %d\n" % (i + 1))
        for m in range(int(mo[i]), int(mo[i + 1])):
            mm = res["mems"][m]
            out.append("MEM START: %d, MEM END: %d BWT START: %d SIZE: %d\n" % (mm["start"], mm["end"], mm["bwt_start"], mm["size"]))
            out.append("Number of unique positions: %d\n" % (int(po[m + 1]) - int(po[m])))
            out.append("".join("%d, " % p for p in res["positions"][int(po[m]):int(po[m + 1])]) + "\n")
        out.append("\n")
    return "".join(out)


def strip_timing(text):
    """drop the non-deterministic trailer (find_mems.cpp:144-145)"""
    i = text.find("\nTotal time for finding all MEMs:")
    assert i >= 0
    tail = text[i:].split("\n")
    assert tail[1].startswith("Total time for finding all MEMs: ") and tail[1].endswith(" seconds")
    assert tail[2].startswith("Total time for all tag queries: ") and tail[2].endswith(" seconds")
    return text[:i]

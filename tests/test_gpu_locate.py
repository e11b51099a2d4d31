"""GPU parity of the locate path (pgx_locate_batch / pgx_locate_next_batch / pgx_decompress_sa) with the oracle and with
the reference's own Locate_* expectation (DA == document array of a brute-force BWT, tests/test_rindex.cpp:103-244)."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W
from test_locate import brute_force_sa

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("txt,rlbwt", [("med_test.txt", "med_test.rl_bwt"), ("x.newline_separated", "x.rl_bwt")])
@pytest.mark.parametrize("encoded", [False, True])
def test_decompress_da_reference_tests(golden, workdir, txt, rlbwt, encoded):
    seqs = W.load_sequences(os.path.join(golden, txt))
    _, da = brute_force_sa(seqs)
    ri = os.path.join(workdir, "gloc_%s_%d.ri" % (rlbwt, encoded))
    P.build_rindex(os.path.join(golden, rlbwt), ri, encoded=encoded)
    for mode in (P.MODE_COMPAT, P.MODE_STRICT):
        idx = P.Index(ri, mode=mode)
        assert np.array_equal(idx.decompress_sa(seq_ids=True), da)
        assert np.array_equal(idx.decompress_sa(), O.RIndex(ri).decompress_sa())
        idx.close()


def test_reference_ri_files(golden):
    for ri, txt in (("bidirectional_test/xy.ri", "bidirectional_test/contigs_xy"), ("two_contig_graph/xy.ri", "two_contig_graph/contigs_XY.txt")):
        seqs = W.load_sequences(os.path.join(golden, txt))
        _, da = brute_force_sa(seqs)
        idx = P.Index(os.path.join(golden, ri))
        assert np.array_equal(idx.decompress_sa(seq_ids=True), da)
        r = O.RIndex(os.path.join(golden, ri))
        sa = r.decompress_sa()
        assert np.array_equal(idx.decompress_sa(), sa)
        # locateNext, incl. the undefined tail of the chain
        probe = np.concatenate([sa, np.array([P.NO_POSITION, 0, int(sa.max()) + 5], dtype=np.uint64)])
        exp = np.array([r.locate_next(int(v)) for v in probe], dtype=np.uint64)
        assert np.array_equal(idx.locate_next_batch(probe), exp)


def test_locate_ranges_all_flag_combinations(golden, workdir):
    rng = np.random.default_rng(31)
    cases = [(os.path.join(golden, "bidirectional_test", "xy.ri"), P.MODE_COMPAT)]  # legacy layout: COMPAT == STRICT
    enc = os.path.join(workdir, "gloc_x_enc.ri")
    P.build_rindex(os.path.join(golden, "x.rl_bwt"), enc, encoded=True)
    cases.append((enc, P.MODE_STRICT))
    for ri, mode in cases:
        r = O.RIndex(ri)
        n = r.n
        sa = r.decompress_sa()
        first = rng.integers(0, n, 3000).astype(np.uint64)
        ln = np.concatenate([rng.integers(0, 30, 2500), rng.integers(30, 3000, 490), rng.integers(3000, n, 10)]).astype(np.uint64)
        last = np.minimum(first + ln, np.uint64(n - 1))
        # empty states (last < first), single positions, the whole BWT
        first = np.concatenate([first, np.array([5, 0, n - 1, 0], dtype=np.uint64)])
        last = np.concatenate([last, np.array([4, 0, n - 1, n - 1], dtype=np.uint64)])
        idx = P.Index(ri, mode=mode)
        ml = np.uint64(r.max_length)
        for flags in (0, P.LOCATE_SEQ_IDS, P.LOCATE_UNIQUE, P.LOCATE_SEQ_IDS | P.LOCATE_UNIQUE):
            off, vals = idx.locate_batch(first, last, flags)
            assert len(off) == len(first) + 1 and off[-1] == len(vals)
            for i in range(len(first)):
                a, b = int(first[i]), int(last[i])
                e = sa[a:b + 1] if b >= a else sa[:0]
                if flags & P.LOCATE_SEQ_IDS:
                    e = e // ml
                if flags & P.LOCATE_UNIQUE:
                    e = np.unique(e)
                assert np.array_equal(vals[int(off[i]):int(off[i + 1])], e), (ri, flags, i, a, b)
        # the oracle's literal locate on a sample of the ranges (sorted unique sequence ids)
        off, vals = idx.locate_batch(first[:300], last[:300], P.LOCATE_SEQ_IDS | P.LOCATE_UNIQUE)
        for i in range(300):
            assert np.array_equal(vals[int(off[i]):int(off[i + 1])], r.locate(int(first[i]), int(last[i]), O.MODE_STRICT if mode == P.MODE_STRICT else O.MODE_COMPAT))
        idx.close()


def test_locate_errors_and_compat_quirk(golden, workdir):
    enc = os.path.join(workdir, "gloc_x_enc2.ri")
    P.build_rindex(os.path.join(golden, "x.rl_bwt"), enc, encoded=True)
    idx = P.Index(enc, mode=P.MODE_COMPAT)
    with pytest.raises(P.PgxError) as e:  # encoded, no N: the reference's run scan is broken (quirk 3)
        idx.locate_batch([0], [5])
    assert e.value.code == P.ERR_UNSUPPORTED
    assert len(idx.decompress_sa()) == idx.info().bwt_size  # decompressSA never scans blocks: fine in COMPAT
    idx.close()
    idx = P.Index(enc, mode=P.MODE_STRICT)
    with pytest.raises(P.PgxError) as e:
        idx.locate_batch([0], [idx.info().bwt_size])
    assert e.value.code == P.ERR_ARG
    off, vals = idx.locate_batch([], [])
    assert list(off) == [0] and len(vals) == 0


def test_locate_on_synthetic_pangenome(workdir):
    # sigma = 6 index with long runs and many sequences: DA from the device == oracle chain; MEM intervals located
    text = os.path.join(workdir, "loc_synth.txt")
    W.synth_pangenome_text(text, base_len=20000, n_hap=4, seed=77)
    ri = W.build_index_from_text(text, workdir, "loc_synth", encoded=True, with_tags=False)[0]
    r = O.RIndex(ri)
    idx = P.Index(ri)
    sa = r.decompress_sa()
    assert np.array_equal(idx.decompress_sa(), sa)
    seqs = W.load_sequences(text)
    total = sum(len(s) + 1 for s in seqs)
    assert total == r.n
    # every suffix position appears exactly once
    ml = r.max_length
    got = np.sort((sa // np.uint64(ml)) * np.uint64(ml) + sa % np.uint64(ml))
    assert len(np.unique(got)) == r.n
    rng = np.random.default_rng(3)
    first = rng.integers(0, r.n, 2000).astype(np.uint64)
    last = np.minimum(first + rng.integers(0, 200, 2000).astype(np.uint64), np.uint64(r.n - 1))
    off, vals = idx.locate_batch(first, last, 0)
    for i in range(0, 2000, 7):
        assert np.array_equal(vals[int(off[i]):int(off[i + 1])], sa[int(first[i]):int(last[i]) + 1])

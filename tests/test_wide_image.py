"""WIDE form of the dense2 / PAIRS images (pgx_image.h "WIDE": BWTs of 2^32 symbols or more; the reference is size_t end to end,
include/pangenome_index/r-index.hpp:118-130).  CPU tier: the wide image of an index is its narrow image re-based -- every header
count of a block plus the base of the block's superblock equals the narrow (absolute) count, everything else is the same bytes --,
for superblocks so small (PGX_SB_SHIFT) that a small index has dozens of them.  The narrow images themselves are checked against
brute force in tests/test_image.py and tests/test_pairs_image.py.  GPU tier: the 64-bit kernels on such images against the oracle."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W
from image_emu import Consts


@pytest.fixture(scope="module")
def wide_case(workdir):
    text = os.path.join(workdir, "wide_case.txt")
    W.synth_pangenome_text(text, base_len=30_000, n_hap=3, seed=77, snp=0.01, indel=0.001, n_runs=3, n_run_len=(20, 400))
    ri = W.build_index_from_text(text, workdir, "wide_case", with_tags=True)
    return ri[0], ri[1], text


@pytest.mark.parametrize("shift,stride", [("2", "96"), ("4", "96"), ("7", "96"), ("3", "64"), ("6", "64")])
def test_wide_images_are_the_narrow_ones_rebased(wide_case, monkeypatch, shift, stride):
    ri, _, _ = wide_case
    monkeypatch.setenv("PGX_PAIRS_STRIDE", stride)  # both strides of the PAIRS blocks, narrow and wide alike
    narrow = P.Index(ri, None, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
    monkeypatch.setenv("PGX_SB_SHIFT", shift)
    wide = P.Index(ri, None, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS | P.MODE_IMAGE_WIDE)
    assert narrow.info().image_wide == 0 and wide.info().image_wide == 1 and wide.info().image_pairs == 1
    cn, cw = Consts(narrow.image_view(6)), Consts(wide.image_view(6))
    assert cn.wide == 0 and cw.wide == 1 and cw.n == cn.n and cw.C == cn.C and cw.ext_tab == cn.ext_tab
    assert list(cw.pair_t2w) == [int(v) for v in cn.pair_t2] == list(cn.pair_t2w)
    # dense2: header dwords 0..4 (A C G T N) are deltas; dword 5 .. 31 (exceptions, sub-block counts, planes) are the same
    bn, bw = narrow.image_view(0).view(np.uint32).reshape(-1, 32), wide.image_view(0).view(np.uint32).reshape(-1, 32)
    sb2 = wide.image_view(22).reshape(-1, 8)
    assert len(bn) == len(bw) and len(sb2) == cw.n_sb2 == ((len(bw) - 1) >> cw.d2_sb_shift) + 1 and 1 < cw.n_sb2 <= 64
    assert np.array_equal(bn[:, 5:], bw[:, 5:])
    which = np.arange(len(bw)) >> cw.d2_sb_shift
    assert np.array_equal(bw[:, :5].astype(np.uint64) + sb2[which, :5], bn[:, :5].astype(np.uint64))
    assert np.array_equal(sb2[:, 5], sb2[:, :5].sum(axis=1)) and not sb2[:, 6:].any()
    assert np.array_equal(sb2[1:, :5], bn[np.arange(1, len(sb2)) << cw.d2_sb_shift, :5])  # a base = the absolute counts at the superblock's first block
    assert np.array_equal(narrow.image_view(15), wide.image_view(15))
    # pairs: the sixteen pair counts are deltas, row sums of the bases are kept next to them; dword 16 .. 31 are the same
    pn, pw = narrow.image_view(20).reshape(-1, 32), wide.image_view(20).reshape(-1, 32)
    pb = wide.image_view(23).reshape(-1, 24)
    assert len(pn) == len(pw) and len(pb) == cw.n_sbp == ((len(pw) - 1) >> cw.pairs_sb_shift) + 1 and 1 < cw.n_sbp <= 64
    assert np.array_equal(pn[:, 16:], pw[:, 16:])
    whichp = np.arange(len(pw)) >> cw.pairs_sb_shift
    assert np.array_equal(pw[:, :16].astype(np.uint64) + pb[whichp, :16], pn[:, :16].astype(np.uint64))
    assert np.array_equal(pb[:, 16:20], pb[:, :16].reshape(-1, 4, 4).sum(axis=2)) and not pb[:, 20:].any()
    assert (pw[:, :16].astype(np.int64) < (1 << 31)).all()
    narrow.close(); wide.close()


def test_wide_is_refused_for_the_other_layouts(wide_case):
    ri, _, _ = wide_case
    for layout in (P.MODE_IMAGE_RL, P.MODE_IMAGE_DENSE):
        with pytest.raises(P.PgxError) as e:
            P.Index(ri, None, mode=P.MODE_COMPAT | layout | P.MODE_IMAGE_WIDE)
        assert e.value.code == P.ERR_ARG
    idx = P.Index(ri, None, mode=P.MODE_COMPAT | P.MODE_IMAGE_WIDE)  # alone: dense2 + pairs, both wide
    assert idx.info().image_kind == P.IMAGE_DENSE2 and idx.info().image_wide == 1 and idx.info().image_pairs == 1
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shift", ["2", "22"])
def test_wide_kernels_against_the_oracle(wide_case, monkeypatch, shift):
    """pgx_find_mems_pairs_kernel<.., WIDE> + pgx_find_mems_kernel<false, 3, ..> (64-bit state, superblock bases in LDS) on a small index
    forced into the wide form, many superblocks (shift 2) and one (shift 22): rank at every position, random extensions, find_mems + tags."""
    ri_path, tags_path, text = wide_case
    monkeypatch.setenv("PGX_SB_SHIFT", shift)
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 100_000, 150, seed=5)
    extra = [b"N" * 150, bytes(seqs[0][-150:]), bytes(seqs[1][:150]), b"acgt" * 30, b"ACGTNACGT" * 10, b""]
    ecat, eoffs = O.pack_reads(extra)
    cat = np.concatenate([cat, ecat]); offs = np.concatenate([offs, eoffs[1:] + offs[-1]])
    for force in (P.MODE_IMAGE_PAIRS, P.MODE_IMAGE_DENSE2):
        idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | force | P.MODE_IMAGE_WIDE)
        assert idx.info().image_wide == 1 and idx.info().image_pairs == (1 if force == P.MODE_IMAGE_PAIRS else 0)
        # primitives, in launches of far more than 256 workgroups and twice each (round 3: a looped rank kernel returned stale counts in
        # workgroups beyond the first wave of the grid, differently from call to call)
        pos = np.arange(0, ri.n + 2, dtype=np.uint64)
        exp_rank = np.array([ri.rank6_true(min(p, ri.n)) for p in range(ri.n + 2)], dtype=np.uint64)
        for _ in range(2):
            assert np.array_equal(idx.rank_batch(pos, true_codes=True), exp_rank)
        rng = np.random.default_rng(11)
        m = 120_000
        iv = np.zeros(m, dtype=P.BIINT_DTYPE)
        iv["forward"] = rng.integers(0, ri.n, m)
        iv["size"] = 1 + (rng.random(m) * np.minimum(ri.n - iv["forward"], 3000)).astype(np.int64)
        iv["reverse"] = rng.integers(0, ri.n, m)
        syms = np.frombuffer(b"ACGTN\n\x00a", dtype=np.uint8)[rng.integers(0, 8, m)]
        fw = rng.integers(0, 2, m).astype(np.uint8)
        got1, got2 = idx.extend_batch(iv, syms, fw), idx.extend_batch(iv, syms, fw)
        assert got1.tobytes() == got2.tobytes()
        for i in range(0, m, 37):
            tri = (int(iv["forward"][i]), int(iv["reverse"][i]), int(iv["size"][i]))
            e = (ri.fwd if fw[i] else ri.bwd)(tri, int(syms[i]))
            assert (int(got1["forward"][i]), int(got1["reverse"][i]), int(got1["size"][i])) == e, (i, tri)
        n_reads = len(offs) - 1
        cnt1, cnt2 = idx.count_batch(cat, offs), idx.count_batch(cat, offs)  # unidirectional search: a loop of LF steps per read
        assert np.array_equal(cnt1, cnt2)
        for i in range(0, n_reads, 53):
            rd = bytes(cat[int(offs[i]):int(offs[i + 1])])
            assert (int(cnt1[i][0]), int(cnt1[i][1])) == ri.count(rd), i
        ro = np.repeat(np.arange(0, n_reads, 25, dtype=np.uint64), 6)  # find_mems_function at six start positions of every 25th read
        xs = np.tile(np.array([0, 1, 17, 60, 129, 149], dtype=np.uint64), len(ro) // 6)
        f1, f2 = idx.find_mems_function_batch(cat, offs, ro, xs, 20, 1), idx.find_mems_function_batch(cat, offs, ro, xs, 20, 1)
        assert all(np.array_equal(a, b) for a, b in zip(f1, f2))
        for q in range(0, len(ro), 41):
            rd = bytes(cat[int(offs[int(ro[q])]):int(offs[int(ro[q]) + 1])])
            nx, mem, ne = ri.find_mems_function(rd, 20, 1, int(xs[q]))
            assert int(f1[0][q]) == nx and int(f1[3][q]) == ne and bool(f1[2][q]) == (mem is not None), q
            if mem is not None:
                assert tuple(int(v) for v in f1[1][q]) == mem
        for min_len, min_occ in ((20, 1), (12, 1), (25, 3), (5, 2), (31, 1)):
            ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=4)
            res = idx.find_mems(cat, offs, min_len, min_occ, tags=True)
            assert idx.find_mems(cat, offs, min_len, min_occ, tags=True)["mems"].tobytes() == res["mems"].tobytes()
            assert np.array_equal(res["mem_offsets"], ref["mem_offsets"]), (force, min_len, min_occ)
            assert res["mems"].tobytes() == ref["mems"].tobytes()
            assert res["n_extensions"] == ref["n_extensions"]
            assert np.array_equal(res["pos_offsets"], ref["pos_offsets"]) and np.array_equal(res["positions"], ref["positions"])
        b = idx.batch(cat, offs)
        b.run(20, 1, P.RUN_TIMING)
        assert (b.timing().pairs_reads != 0) == (force == P.MODE_IMAGE_PAIRS) and b.timing().seed_depth > 0
        b.free()
        idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("wide,stride", [(0, "96"), (1, "96"), (0, "64"), (1, "64")])
def test_cooperative_line_fetches_against_the_oracle(wide_case, monkeypatch, wide, stride):
    """pgx_find_mems_pairs_kernel<.., COOP>: the wave fetches the 64 block lines of its lanes together through LDS (global_load_lds), the variant
    for PAIRS images beyond the reach of the address-translation caches -- forced here on a small index, narrow and wide, with reads that
    meet flagged blocks, heavy reads and idle lanes in the mix"""
    ri_path, tags_path, text = wide_case
    monkeypatch.setenv("PGX_SB_SHIFT", "3")
    monkeypatch.setenv("PGX_FM_COOP", "1")
    monkeypatch.setenv("PGX_PAIRS_STRIDE", stride)
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 60_000, 150, seed=9)
    extra = [bytes(seqs[0][-150:]), bytes(seqs[1][:150]), bytes(seqs[2][-40:]), b"ACGT" * 30, b"", b"A"]
    ecat, eoffs = O.pack_reads(extra)
    cat = np.concatenate([cat, ecat]); offs = np.concatenate([offs, eoffs[1:] + offs[-1]])
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS | (P.MODE_IMAGE_WIDE if wide else 0))
    for min_len, min_occ in ((20, 1), (12, 1), (25, 3), (31, 1)):
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=4)
        b = idx.batch(cat, offs)
        for _ in range(2):
            b.run(min_len, min_occ, P.RUN_TAGS | P.RUN_TIMING)
            assert b.timing().pairs_reads == 3  # the cooperative variant ran
            res = b.result()
            assert np.array_equal(res["mem_offsets"], ref["mem_offsets"]), (wide, min_len, min_occ)
            assert res["mems"].tobytes() == ref["mems"].tobytes()
            assert res["n_extensions"] == ref["n_extensions"]
            assert np.array_equal(res["pos_offsets"], ref["pos_offsets"]) and np.array_equal(res["positions"], ref["positions"])
        b.free()
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [1, 0])
def test_every_extension_through_the_other_image(workdir, monkeypatch, wide):
    """The product twin of the round-3 rank primitive that looped over probes (DESIGN.md "stale counts"): the rolled two-probe loop the pairs kernel
    takes when its own block cannot answer (pgx_kernels.hip, `bail`).  A text with an N at every twentieth position flags EVERY block of the PAIRS
    image (some position of each has N as first or second symbol), so every extension behind a stage's first goes through that loop -- 64-bit
    (pgx_dense2w_rank, superblock bases) and narrow --, in a launch of hundreds of workgroups, every run twice, against the oracle."""
    rng = np.random.default_rng(123)
    text = os.path.join(workdir, "n_rich.txt")
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = acgt[rng.integers(0, 4, 40_000)].copy()
    with open(text, "wb") as f:
        for h in range(3):
            s = base.copy()
            m = rng.random(len(s)) < 0.01
            s[m] = acgt[rng.integers(0, 4, int(m.sum()))]
            s[rng.random(len(s)) < 0.05] = ord("N")
            f.write(s.tobytes() + b"\n")
            f.write(W._COMP[s[::-1]].tobytes() + b"\n")
    ri_path, tags_path = W.build_index_from_text(text, workdir, "n_rich")[:2]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    assert ri.sigma == 6
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 60_000, 150, seed=4)
    isn = cat == ord("N")
    cat = cat.copy()
    cat[isn] = acgt[rng.integers(0, 4, int(isn.sum()))]  # pure A C G T: every read stays with the pairs kernel
    monkeypatch.setenv("PGX_SEED_K", "6")
    monkeypatch.setenv("PGX_SB_SHIFT", "3")
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS | (P.MODE_IMAGE_WIDE if wide else 0))
    assert idx.info().image_pairs == 1 and idx.info().image_wide == wide
    pv = idx.image_view(20).view(np.uint32).reshape(-1, 32)
    assert (pv[:-1, 16] >> 31).all()  # every block that holds a position is flagged
    for min_len, min_occ in ((8, 1), (11, 2)):
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
        b = idx.batch(cat, offs)
        for _ in range(2):
            b.run(min_len, min_occ, P.RUN_TAGS | P.RUN_TIMING)
            t = b.timing()
            assert t.pairs_reads == (4 if (not wide and min_occ <= 1) else 2) and t.pairs_other_steps > 10 * (len(offs) - 1)  # the two-step kernel ran, and took its extensions through the other image
            res = b.result()
            assert np.array_equal(res["mem_offsets"], ref["mem_offsets"]), (wide, min_len, min_occ)
            assert res["mems"].tobytes() == ref["mems"].tobytes()
            assert res["n_extensions"] == ref["n_extensions"]
            assert np.array_equal(res["pos_offsets"], ref["pos_offsets"]) and np.array_equal(res["positions"], ref["positions"])
        b.free()
    idx.close()

"""Full-size parity of the sigma = 6 synthetic pangenome workload inside the -m gpu tier: SURVEY 8d
config-3 recipe at 1/10 scale (4 Mbp x 8 haplotypes x 2 strands, n = 64 M, r = 6.3 M), 1 M synthetic 150-bp reads,
min_len 20 -- about 1.9 M MEMs, 23 M positions, 196 M extensions -- every MEM, run count and position against the CPU
oracle, under every layout of the device rank image.  (The same check at chr22 scale, n = 640 M, is
scripts/parity_full_synth.py; its output is committed under profiles/.)"""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W

pytestmark = pytest.mark.gpu


def test_synth_pangenome_one_million_reads(workdir):
    text = os.path.join(workdir, "full_synth.txt")
    W.synth_pangenome_text(text, base_len=4_000_000)
    ri_path, tags_path = W.build_index_from_text(text, workdir, "full_synth")[:2]
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 1_000_000, 150, seed=42 + 3)
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    assert ri.sigma == 6 and ri.n > 60_000_000
    ref = O.find_mems_batch(ri, tags, cat, offs, 20, 1, threads=O.lib().orc_max_threads())
    assert len(ref["mems"]) > 1_500_000 and len(ref["positions"]) > 10_000_000 and ref["n_extensions"] > 150_000_000
    # the automatic layout (64-byte dense blocks + the two-step pairs image + seed table), dense2 + pairs (the automatic layout of longer BWTs),
    # the 64-byte blocks alone, dense2 alone, run-length blocks
    for force in (0, P.MODE_IMAGE_PAIRS, P.MODE_IMAGE_DENSE, P.MODE_IMAGE_DENSE2, P.MODE_IMAGE_RL):
        idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | force)
        assert force or idx.info().image_kind == P.IMAGE_DENSE
        assert idx.info().image_pairs == (1 if force in (0, P.MODE_IMAGE_PAIRS) else 0)
        res = idx.find_mems(cat, offs, 20, 1, tags=True)
        assert np.array_equal(res["mem_offsets"], ref["mem_offsets"])
        assert res["mems"].tobytes() == ref["mems"].tobytes()
        assert res["n_extensions"] == ref["n_extensions"]
        assert np.array_equal(res["tag_run_counts"], ref["tag_run_counts"])
        assert np.array_equal(res["pos_offsets"], ref["pos_offsets"])
        assert np.array_equal(res["positions"], ref["positions"])
        assert res["n_tag_overflow"] == ref["n_tag_overflow"]
        idx.close()


def test_chr22_scale_automatic_layout(workdir):
    """BASELINE configs[2] at its own size (the bench default; reference unit: src/find_mems.cpp:94-139): n = 640 M, automatic layout
    (dense2 + two-step pairs image, seed table of depth 16 = 64 GiB + the depth-10 table + end table), 1 M reads bit-identical to the
    oracle; then the bench batch shape -- 10 M reads in ONE chunk -- whose first 1 M reads are those reads: same results inside the bigger batch."""
    text = os.path.join(workdir, "chr22_synth.txt")
    W.synth_pangenome_text(text, base_len=40_000_000, n_hap=8, seed=45)  # bench.py's chr22 workload
    ri_path, tags_path = W.build_index_from_text(text, workdir, "chr22_synth")[:2]
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 10_000_000, 150, seed=42 + 3)
    del seqs
    n1 = 1_000_000
    cat1, offs1 = cat[: n1 * 150], offs[: n1 + 1]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    assert ri.sigma == 6 and ri.n > 600_000_000
    ref = O.find_mems_batch(ri, tags, cat1, offs1, 20, 1, threads=O.lib().orc_max_threads())
    del ri, tags
    assert len(ref["mems"]) > 1_500_000 and ref["n_extensions"] > 150_000_000
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT)
    info = idx.info()
    assert info.image_kind == P.IMAGE_DENSE2 and info.image_pairs == 1 and not info.image_in_lds
    b1 = idx.batch(cat1, offs1)
    b1.run(20, 1, P.RUN_TAGS | P.RUN_TIMING)
    t = b1.timing()
    assert t.pairs_reads == 4 and t.main_lines > 0 and t.main_seed_loads > 0  # the two-step kernel behind the seed table ran, reads packed in LDS, narrow forward stages through the text
    assert t.seed_depth == 16 and t.find_mems_launches == 1
    res = b1.result()
    b1.free()

    def same(res, m, npos):
        assert np.array_equal(res["mem_offsets"][: n1 + 1], ref["mem_offsets"])
        assert res["mems"][:m].tobytes() == ref["mems"].tobytes()
        assert np.array_equal(res["tag_run_counts"][:m], ref["tag_run_counts"])
        assert np.array_equal(res["pos_offsets"][: m + 1], ref["pos_offsets"])
        assert np.array_equal(res["positions"][:npos], ref["positions"])

    m, npos = len(ref["mems"]), len(ref["positions"])
    same(res, m, npos)
    assert res["n_extensions"] == ref["n_extensions"] and res["n_tag_overflow"] == ref["n_tag_overflow"]
    assert len(res["mems"]) == m and len(res["positions"]) == npos
    del res
    b10 = idx.batch(cat, offs)
    b10.run(20, 1, P.RUN_TAGS | P.RUN_TIMING)
    assert b10.timing().find_mems_launches == 1 and b10.timing().seed_depth == 16  # one chunk, as bench.py measures it
    res10 = b10.result()
    b10.free()
    same(res10, m, npos)
    assert len(res10["mem_offsets"]) == 10_000_001 and res10["n_extensions"] > 9 * ref["n_extensions"]
    idx.close()


def _whole_genome_case(workdir, name, chroms, base_len, haps, n_reads, n1, full):
    """the body of the whole-genome-scale test; full = False: the same recipe on a small collection forced into the 64-bit form with
    superblocks of a few blocks (PGX_SB_SHIFT set by the caller), so that the recipe itself -- border reads included -- is exercised in seconds"""
    from image_emu import Consts

    texts = W.synth_chromosome_texts(workdir, name, chroms, base_len, haps, seed=45, n_runs=2, n_run_len=(1000, 10000) if full else (50, 400))
    ri_path, tags_path = W.build_index_from_texts(texts, workdir, name)[:2]
    seqs = []
    for t in texts:
        seqs += W.load_sequences(t)
    assert len(seqs) == 2 * chroms * haps
    cat, offs = W.sample_reads(seqs, n_reads, 150, seed=42 + 5)
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    assert ri.sigma == 6
    if full:
        assert ri.n == 4_351_996_034 and ri.n > 1 << 32

    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | (0 if full else P.MODE_IMAGE_WIDE))
    info = idx.info()
    assert info.image_kind == P.IMAGE_DENSE2 and info.image_pairs == 1 and info.image_wide == 1 and info.pairs_stride == 64 and not info.image_in_lds
    c = Consts(idx.image_view(6))
    assert c.wide == 1 and 2 <= c.n_sb2 <= 64 and 2 <= c.n_sbp <= 64
    # reads whose search ends in an interval next to a superblock border: the text at the suffix array's values there (packed as
    # sequence * max_length + offset, src/r-index.cpp:1300-1343)
    borders = [k * (384 << c.d2_sb_shift) for k in range(1, c.n_sb2)] + [k * (c.pairs_stride << c.pairs_sb_shift) for k in range(1, c.n_sbp)]
    borders = [b for b in borders if 100 < b < ri.n - 100]
    assert len(borders) >= 4
    extra, near = [], 0
    ml = ri.max_length
    for b in borders:
        sa = ri.locate_sa(b - 24, b + 24)
        for i, v in enumerate(sa):
            s, o = int(v) // ml, int(v) % ml
            rd = bytes(seqs[s][o:o + 150])
            if len(rd) == 150 and b"N" not in rd:
                lo, hi = ri.count(rd)
                assert lo <= b - 24 + i <= hi
                near += 1 if lo <= b <= hi + 1 else 0
                extra.append(rd)
    assert near >= len(borders) // 2  # final intervals that touch a border itself
    q = len(seqs) // 2
    extra += [b"N" * 150, bytes(seqs[0][-150:]), bytes(seqs[1][:150]), bytes(seqs[-1][-150:]), bytes(seqs[q][:150]), b"acgt" * 30, b"ACGTNACGT" * 10, b"",
              bytes(seqs[7][1000:1100]), bytes(seqs[q + 1][-40:])]
    del seqs
    ecat, eoffs = O.pack_reads(extra)
    cat1 = np.concatenate([cat[: n1 * 150], ecat])
    offs1 = np.concatenate([offs[: n1 + 1], eoffs[1:] + offs[n1]])
    ref = O.find_mems_batch(ri, tags, cat1, offs1, 20, 1, threads=O.lib().orc_max_threads())
    del ri, tags
    assert len(ref["mems"]) > 1.5 * n1 and ref["n_extensions"] > 150 * n1

    b1 = idx.batch(cat1, offs1)
    b1.run(20, 1, P.RUN_TAGS | P.RUN_TIMING)
    t = b1.timing()
    assert t.find_mems_launches == 1 and t.main_lines > 0 and t.seed_depth > 0
    if full:
        assert t.pairs_reads == 3 and t.seed_depth == 16  # cooperative line fetches + packed reads, seed table of depth 16 (64 GiB, 40-bit fields)
    else:
        assert t.pairs_reads == 2
    res = b1.result()
    b1.free()
    assert np.array_equal(res["mem_offsets"], ref["mem_offsets"])
    assert res["mems"].tobytes() == ref["mems"].tobytes()
    assert res["n_extensions"] == ref["n_extensions"] and res["n_tag_overflow"] == ref["n_tag_overflow"]
    assert np.array_equal(res["tag_run_counts"], ref["tag_run_counts"])
    assert np.array_equal(res["pos_offsets"], ref["pos_offsets"])
    assert np.array_equal(res["positions"], ref["positions"])
    if full:
        assert int(res["mems"]["bwt_start"].max()) > 1 << 32  # (64-bit coordinates in the output)
    del res
    b10 = idx.batch(cat, offs)
    b10.run(20, 1, P.RUN_TAGS | P.RUN_TIMING)
    assert b10.timing().find_mems_launches == 1 and b10.timing().pairs_reads == (3 if full else 2)  # one chunk, as bench.py --workload wg measures it
    res10 = b10.result()
    b10.free()
    m = int(ref["mem_offsets"][n1])
    npos = int(ref["pos_offsets"][m])
    assert np.array_equal(res10["mem_offsets"][: n1 + 1], ref["mem_offsets"][: n1 + 1])
    assert res10["mems"][:m].tobytes() == ref["mems"][:m].tobytes()
    assert np.array_equal(res10["tag_run_counts"][:m], ref["tag_run_counts"][:m])
    assert np.array_equal(res10["pos_offsets"][: m + 1], ref["pos_offsets"][: m + 1])
    assert np.array_equal(res10["positions"][:npos], ref["positions"][:npos])
    assert len(res10["mem_offsets"]) == n_reads + 1
    idx.close()
    for p in texts:
        os.remove(p)


def test_whole_genome_recipe_on_a_small_collection(workdir, monkeypatch):
    """the recipe of the next test -- several chromosome texts merged by pgx_build_index_from_texts, reads at the superblock borders of both
    64-bit images, a small batch and the prefix of a big one -- on 3 chromosomes x 60 kbp x 4 haplotypes, forced into the 64-bit form with
    superblocks of 2^7 dense2 blocks and 2^9 PAIRS blocks (about 30 and 44 of them)"""
    monkeypatch.setenv("PGX_SB_SHIFT", "7")
    _whole_genome_case(workdir, "wg_small", 3, 60_000, 4, 200_000, 20_000, full=False)


def test_whole_genome_scale_wide_layout(workdir):
    """BASELINE configs[4]'s single-GPU ingredient at its own size (the reference is size_t end to end: include/pangenome_index/r-index.hpp:118-130,
    src/r-index.cpp:713-756): one merged index of n = 4 351 996 034 > 2^32 symbols (bench.py --workload wg: 8 chromosomes x 8.5 Mbp x 32 haplotypes
    x 2 strands, 512 sequences) built by pgx_build_index_from_texts; the automatic layout must be the 64-bit one (WIDE dense2 + WIDE PAIRS at
    stride 64 behind cooperative line fetches, seed table of depth 16 with 40-bit fields).  100 k sampled reads plus reads that end / start a
    sequence, an all-N read, lower case, and reads whose BWT intervals lie across the superblock borders of both images (their 64-bit bases
    change there) bit-identical to the oracle incl. tags and n_extensions; then the bench batch shape -- 10 M reads in one chunk -- whose first
    100 k reads give the same bytes."""
    _whole_genome_case(workdir, "wg_8_8500000_32", 8, 8_500_000, 32, 10_000_000, 100_000, full=True)

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
    (dense2 + two-step pairs image, seed table of depth 15 = 16 GiB + the depth-10 table + end table), 1 M reads bit-identical to the
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
    assert t.pairs_reads == 2 and t.main_lines > 0 and t.main_seed_loads > 0  # the two-step kernel behind the seed table ran, reads packed in LDS
    assert t.seed_depth == 15 and t.find_mems_launches == 1
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
    assert b10.timing().find_mems_launches == 1 and b10.timing().seed_depth == 15  # one chunk, as bench.py measures it
    res10 = b10.result()
    b10.free()
    same(res10, m, npos)
    assert len(res10["mem_offsets"]) == 10_000_001 and res10["n_extensions"] > 9 * ref["n_extensions"]
    idx.close()

"""Heavy reads: a read whose suffix ends a sequence costs ~len^2/2 extensions in one chain (pattern[len] = 0 acts as the
endmarker).  The find_mems kernel hands the rest of such a read to pgx_find_mems_heavy_kernel, which
evaluates every remaining start position at once; results and the extension count must not change."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W

pytestmark = pytest.mark.gpu


def _same(res, ref):
    assert np.array_equal(res["mem_offsets"], ref["mem_offsets"])
    assert res["mems"].tobytes() == ref["mems"].tobytes()
    assert res["n_extensions"] == ref["n_extensions"]
    assert np.array_equal(res["tag_run_counts"], ref["tag_run_counts"])
    assert np.array_equal(res["pos_offsets"], ref["pos_offsets"])
    assert np.array_equal(res["positions"], ref["positions"])


def _run(idx, cat, offs, min_len, min_occ):
    b = idx.batch(cat, offs)
    try:
        b.run(min_len, min_occ, P.RUN_TAGS | P.RUN_TIMING)
        return b.result(), b.timing().heavy_reads
    finally:
        b.free()


def test_reads_from_sequence_ends(workdir):
    text = os.path.join(workdir, "heavy.txt")
    W.synth_pangenome_text(text, base_len=30000, n_hap=2, seed=5, n_runs=1, n_run_len=(50, 200))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "heavy")[:2]
    seqs = W.load_sequences(text)
    ends = [bytes(s[-150:]) for s in seqs] + [bytes(s[-97:]) for s in seqs[:2]]
    cat0, offs0 = W.sample_reads(seqs, 3000, 150, seed=9)
    normal = [bytes(cat0[int(offs0[i]):int(offs0[i + 1])]) for i in range(3000)]
    reads = normal[:1500] + ends + normal[1500:]
    cat, offs = O.pack_reads(reads)
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    cost = [ri.find_all_mems(r, 20, 1, with_ext=True)[1] for r in ends]
    assert max(cost) > 8000 and np.median([ri.find_all_mems(r, 20, 1, with_ext=True)[1] for r in normal[:200]]) < 600
    ref = O.find_mems_batch(ri, tags, cat, offs, 20, 1, threads=O.lib().orc_max_threads())
    for force in (P.MODE_IMAGE_DENSE, P.MODE_IMAGE_RL, P.MODE_IMAGE_DENSE2):
        idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | force)
        res, heavy = _run(idx, cat, offs, 20, 1)
        _same(res, ref)
        assert heavy >= sum(1 for c in cost if c > 4096), (heavy, cost)
        idx.close()


@pytest.mark.parametrize("threshold", ["24", "0"])
def test_every_read_through_the_heavy_kernel(x_index, xy_paths, golden, monkeypatch, threshold):
    """threshold 24: practically every read is handed on after its first start positions; 0 switches the path off"""
    monkeypatch.setenv("PGX_FM_HEAVY_EXT", threshold)
    cases = [(x_index, O.TAGS_COMPACT, "x.newline_separated", P.MODE_COMPAT | P.MODE_IMAGE_DENSE, O.MODE_COMPAT),
             (x_index, O.TAGS_COMPACT, "x.newline_separated", P.MODE_COMPAT | P.MODE_IMAGE_RL, O.MODE_COMPAT),
             (xy_paths, O.TAGS_BYTECODE, "bidirectional_test/contigs_xy", P.MODE_COMPAT, O.MODE_COMPAT),
             (xy_paths, O.TAGS_BYTECODE, "bidirectional_test/contigs_xy", P.MODE_STRICT, O.MODE_STRICT)]
    for (ri_path, tags_path), fmt, text, mode, omode in cases:
        ri, tags = O.RIndex(ri_path), O.Tags(tags_path, fmt)
        seqs = W.load_sequences(os.path.join(golden, text))
        cat, offs = W.sample_reads(seqs, 6000, 150, seed=51)
        idx = P.Index(ri_path, tags_path, mode=mode)
        for min_len, min_occ in [(5, 1), (10, 1), (0, 1), (12, 3)]:
            ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, mode=omode, threads=O.lib().orc_max_threads())
            res, heavy = _run(idx, cat, offs, min_len, min_occ)
            _same(res, ref)
            assert (heavy > 1000) if threshold != "0" else (heavy == 0), heavy
        idx.close()

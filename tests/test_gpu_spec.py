"""Speculative sizing of pgx_batch_run (pgx_runtime.hip): a run whose predecessor on the same batch had the same shape takes its
buffer sizes from that run, keeps every count on the device and synchronises once at the end; when a capacity turns out too
small the run is repeated with exact sizes.  Either way the results are those of the oracle."""
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
    assert res["n_tag_overflow"] == ref["n_tag_overflow"]


@pytest.fixture(scope="module")
def pan(workdir):
    """a pangenome with short N runs and a tag array with one run per BWT position: a read inside an N run has an SA interval of
    a few thousand positions = a tag query of a few thousand runs (the large sort path), and all such reads ask the same query"""
    text = os.path.join(workdir, "specpan.txt")
    W.synth_pangenome_text(text, base_len=150000, n_hap=4, seed=41, n_runs=3, n_run_len=(400, 700))
    ri_path, _, rl = W.build_index_from_text(text, workdir, "specpan", with_tags=False)
    ri = O.RIndex(ri_path)
    vals = ((np.arange(ri.n, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(3000) + np.uint64(1)) << np.uint64(11))
    tags_path = os.path.join(workdir, "specpan.perpos.tags")
    P.write_compact_tags(tags_path, vals, np.ones(ri.n, dtype=np.uint64))
    return ri_path, tags_path, W.load_sequences(text), ri, O.Tags(tags_path, O.TAGS_COMPACT)


def _reads(seqs, n, seed, n_nreads=300):
    cat, offs = W.sample_reads(seqs, n - n_nreads, 150, seed=seed)
    cat = np.concatenate([cat, np.full(n_nreads * 150, ord("N"), dtype=np.uint8)])  # reads inside N runs
    return cat, np.arange(n + 1, dtype=np.uint64) * np.uint64(150)


@pytest.mark.parametrize("force", [0, P.MODE_IMAGE_DENSE, P.MODE_IMAGE_RL])
def test_repeated_runs_are_speculative_and_exact(pan, force):
    ri_path, tags_path, seqs, ri, tags = pan
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | force)
    cat, offs = _reads(seqs, 30000, 3)
    ref = O.find_mems_batch(ri, tags, cat, offs, 20, 1, threads=O.lib().orc_max_threads())
    big = ref["tag_run_counts"][ref["tag_run_counts"] > 2048]
    assert len(big) >= 300 and int(big.max()) <= 16384  # the large path and its device-side grouping of identical queries are exercised
    b = idx.batch(cat, offs)
    for k in range(4):
        b.run(20, 1, P.RUN_TAGS | P.RUN_TIMING)
        _same(b.result(), ref)
        assert b.counts() == (len(ref["mems"]), len(ref["positions"]), ref["n_extensions"])
    assert b.spec_stats() == (3, 0)  # every run after the first: no mid-run read-back, nothing repeated
    # without tags, and with other parameters: the shape changes, the first such run is exact again
    b.run(20, 1, 0)
    b.run(20, 1, 0)
    assert b.result()["mems"].tobytes() == ref["mems"].tobytes() and b.spec_stats() == (4, 0)
    ref2 = O.find_mems_batch(ri, tags, cat, offs, 12, 2, threads=O.lib().orc_max_threads())
    for k in range(2):
        b.run(12, 2, P.RUN_TAGS)
        _same(b.result(), ref2)
    assert b.spec_stats() == (5, 0)
    b.free()
    idx.close()


def test_capacity_too_small_falls_back(pan):
    """same number of reads, same parameters, but the new reads produce far more MEMs / positions than the sizes taken from the
    previous run allow: the speculative run aborts on the device and is repeated exactly"""
    ri_path, tags_path, seqs, ri, tags = pan
    idx = P.Index(ri_path, tags_path)
    n = 20000
    rng = np.random.default_rng(5)
    junk = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n * 150)]  # random reads: hardly any MEM of 20
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)
    real, _ = _reads(seqs, n, 9)
    ref_junk = O.find_mems_batch(ri, tags, junk, offs, 20, 1, threads=O.lib().orc_max_threads())
    ref_real = O.find_mems_batch(ri, tags, real, offs, 20, 1, threads=O.lib().orc_max_threads())
    assert len(ref_real["mems"]) > 4 * len(ref_junk["mems"]) + 1000
    b = idx.batch(junk, offs)
    b.run(20, 1, P.RUN_TAGS)
    _same(b.result(), ref_junk)
    b.upload(real, offs)
    b.run(20, 1, P.RUN_TAGS)  # speculative with the junk batch's sizes -> abort -> exact
    _same(b.result(), ref_real)
    assert b.spec_stats() == (1, 1)
    b.run(20, 1, P.RUN_TAGS)  # now the sizes fit
    _same(b.result(), ref_real)
    b.upload(junk, offs)
    b.run(20, 1, P.RUN_TAGS)  # fewer than predicted: fits as well
    _same(b.result(), ref_junk)
    assert b.spec_stats() == (3, 1)
    b.free()
    idx.close()


def test_switched_off_and_chunked(pan, monkeypatch):
    ri_path, tags_path, seqs, ri, tags = pan
    idx = P.Index(ri_path, tags_path)
    cat, offs = _reads(seqs, 12000, 4)
    ref = O.find_mems_batch(ri, tags, cat, offs, 20, 1, threads=O.lib().orc_max_threads())
    monkeypatch.setenv("PGX_SPEC", "0")
    b = idx.batch(cat, offs)
    for k in range(3):
        b.run(20, 1, P.RUN_TAGS)
        _same(b.result(), ref)
    assert b.spec_stats() == (0, 0)
    monkeypatch.delenv("PGX_SPEC")
    monkeypatch.setenv("PGX_SLOT_BUDGET_MB", "8")  # several chunks of reads: every chunk is sized exactly
    for k in range(2):
        b.run(20, 1, P.RUN_TAGS)
        _same(b.result(), ref)
    assert b.spec_stats() == (0, 0)
    b.free()
    idx.close()


def test_mem_slot_arena_and_its_fallback(monkeypatch, workdir):
    """The fifth and later MEMs of a read go to an arena (an extent reserved when the fifth is emitted) instead of a worst-case region of 131 slots
    per 150-bp read; an arena that proves too small makes the chunk run again in the worst-case layout.  Same bytes either way, also for reads the
    pairs kernel hands on after their fifth MEM and for heavy reads."""
    import pgx_workload as W

    text = os.path.join(workdir, "arena.txt")
    W.synth_pangenome_text(text, base_len=20_000, n_hap=3, seed=91, snp=0.02, indel=0.002, n_runs=2, n_run_len=(10, 100))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "arena")[:2]
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 30_000, 150, seed=4, sub_rate=0.06)  # many mismatches, min_len 8: most reads have more than four MEMs
    extra = [bytes(seqs[0][-150:]), bytes(seqs[1][-150:]), b"N" * 150, b"ACGT" * 37]
    ecat, eoffs = O.pack_reads(extra)
    cat = np.concatenate([cat, ecat]); offs = np.concatenate([offs, eoffs[1:] + offs[-1]])
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    ref = O.find_mems_batch(ri, tags, cat, offs, 8, 1, threads=4)
    per_read = np.diff(ref["mem_offsets"])
    assert (per_read > 4).mean() > 0.5
    for force in (0, P.MODE_IMAGE_PAIRS, P.MODE_IMAGE_RL):
        idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | force)
        # (default arena: eight slots per read at first, then what the last run asked for -- these reads ask for ~100 each, so the first run repeats)
        for env, launches in (({}, None), ({"PGX_SLOT_ARENA": "0"}, 1), ({"PGX_SLOT_ARENA_CAP": "1000"}, 2), ({"PGX_SLOT_ARENA_CAP": "1000", "PGX_SPEC": "0"}, 2)):
            for k in ("PGX_SLOT_ARENA", "PGX_SLOT_ARENA_CAP", "PGX_SPEC"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            b = idx.batch(cat, offs)
            for rep in range(2):  # the second run is sized speculatively (arena from the first)
                b.run(8, 1, P.RUN_TAGS | P.RUN_TIMING)
                res = b.result()
                assert np.array_equal(res["mem_offsets"], ref["mem_offsets"]), (force, env, rep)
                assert res["mems"].tobytes() == ref["mems"].tobytes()
                assert res["n_extensions"] == ref["n_extensions"]
                assert np.array_equal(res["positions"], ref["positions"])
                if launches is not None and (rep == 0 or "PGX_SPEC" in env):
                    assert b.timing().find_mems_launches == launches, (force, env, rep, b.timing().find_mems_launches)
                if launches is None and rep == 1:
                    assert b.timing().find_mems_launches == 1 and b.spec_stats()[1] == 0  # sized from the first run's demand: one launch, nothing repeated
            b.free()
        idx.close()

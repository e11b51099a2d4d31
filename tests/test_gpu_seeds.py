"""k-mer seed table of the dense find_mems kernels (pgx_kernels.hip "k-mer seeds"): results -- MEMs, tag positions and the exact
extension count -- must not depend on the table or on its depth K.  The table is built per device image, so PGX_SEED_K is set
before the index is opened; K = 0 runs the stepwise kernels."""
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


@pytest.fixture(scope="module")
def pan(workdir):
    text = os.path.join(workdir, "seedpan.txt")
    W.synth_pangenome_text(text, base_len=120000, n_hap=4, seed=31, n_runs=3, n_run_len=(100, 3000))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "seedpan")[:2]
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 30000, 150, seed=77)
    # extra reads: bytes outside ACGT inside seed windows, windows at the very start / end of a read, short reads
    rng = np.random.default_rng(3)
    extra = []
    for _ in range(400):
        s = seqs[int(rng.integers(0, len(seqs)))]
        ln = int(rng.integers(1, 200))
        a = int(rng.integers(0, len(s) - ln))
        r = bytearray(bytes(s[a:a + ln]))
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, ln))] = int(rng.choice(np.frombuffer(b"Nacgt\x00$", dtype=np.uint8)))
        extra.append(bytes(r))
    ecat, eoffs = O.pack_reads(extra)
    cat = np.concatenate([cat, ecat])
    offs = np.concatenate([offs, eoffs[1:] + offs[-1]])
    return ri_path, tags_path, cat, offs


_REF = {}  # the oracle's answers, computed once per (mode, min_len, min_occ): they do not depend on the seed table under test


@pytest.mark.parametrize("seed_k", ["0", "3", "8", "11", None])
def test_results_do_not_depend_on_the_seed_table(pan, monkeypatch, seed_k):
    ri_path, tags_path, cat, offs = pan
    if seed_k is None:
        monkeypatch.delenv("PGX_SEED_K", raising=False)  # the automatic depth
    else:
        monkeypatch.setenv("PGX_SEED_K", seed_k)
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    for mode, omode in ((P.MODE_COMPAT | P.MODE_IMAGE_DENSE, O.MODE_COMPAT), (P.MODE_STRICT | P.MODE_IMAGE_DENSE, O.MODE_STRICT),
                        (P.MODE_COMPAT | P.MODE_IMAGE_DENSE2, O.MODE_COMPAT), (P.MODE_STRICT | P.MODE_IMAGE_DENSE2, O.MODE_STRICT)):
        idx = P.Index(ri_path, tags_path, mode=mode)
        assert not idx.info().image_in_lds
        for min_len, min_occ in [(20, 1), (8, 1), (11, 1), (12, 1), (3, 1), (20, 2), (25, 9), (20, 0), (40, 1)]:
            key = (omode, min_len, min_occ)
            if key not in _REF:
                _REF[key] = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, mode=omode, threads=O.lib().orc_max_threads())
            _same(idx.find_mems(cat, offs, min_len, min_occ, tags=True), _REF[key])
        idx.close()


def test_seeds_with_wide_state_and_heavy_reads(pan, monkeypatch):
    """the 64-bit kernels (PGX_FM_NARROW=0) and the heavy-read hand-off see the same table"""
    ri_path, tags_path, cat, offs = pan
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    ref = O.find_mems_batch(ri, tags, cat, offs, 20, 1, threads=O.lib().orc_max_threads())
    for env in ({"PGX_FM_NARROW": "0"}, {"PGX_FM_HEAVY_EXT": "40"}, {"PGX_FM_NARROW": "0", "PGX_FM_HEAVY_EXT": "40", "PGX_SEED_K": "9"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for force in (P.MODE_IMAGE_DENSE, P.MODE_IMAGE_DENSE2):
            idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | force)
            _same(idx.find_mems(cat, offs, 20, 1, tags=True), ref)
            idx.close()
        for k in env:
            monkeypatch.delenv(k)


def test_seeds_on_a_no_n_index_in_compat(workdir, golden, monkeypatch):
    """sigma = 5 (quirk 1: every T kills the interval, reverse coordinates are junk): the table is built with the same tables,
    so its entries carry the quirk values; contigs_xy is too small for the automatic table, so the depth is forced and the
    image kept out of LDS by repeating the text"""
    text = os.path.join(workdir, "non.txt")
    seqs = W.load_sequences(os.path.join(golden, "bidirectional_test", "contigs_xy"))
    rng = np.random.default_rng(1)
    with open(text, "wb") as f:
        for rep in range(30):
            for s in seqs:
                v = np.array(s)
                m = rng.random(len(v)) < 0.02
                v[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
                f.write(v.tobytes() + b"\n")
    ri_path, tags_path = W.build_index_from_text(text, workdir, "non")[:2]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    assert ri.sigma == 5
    cat, offs = W.sample_reads(W.load_sequences(text), 20000, 100, seed=12)
    for seed_k, force in (("0", P.MODE_IMAGE_DENSE), ("6", P.MODE_IMAGE_DENSE), ("0", P.MODE_IMAGE_DENSE2), ("6", P.MODE_IMAGE_DENSE2)):
        monkeypatch.setenv("PGX_SEED_K", seed_k)
        idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | force)
        assert not idx.info().image_in_lds
        for min_len, min_occ in [(6, 1), (10, 1), (8, 3)]:
            ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
            _same(idx.find_mems(cat, offs, min_len, min_occ, tags=True), ref)
        idx.close()

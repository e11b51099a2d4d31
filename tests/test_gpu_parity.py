"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact.  Needs an MI355X."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W

pytestmark = pytest.mark.gpu

SWEEP = [(5, 1), (10, 1), (20, 1), (20, 5)]


def _assert_same(res, ref, tags):
    assert np.array_equal(res["mem_offsets"], ref["mem_offsets"])
    assert res["mems"].tobytes() == ref["mems"].tobytes()
    assert res["n_extensions"] == ref["n_extensions"]
    if tags:
        assert np.array_equal(res["tag_run_counts"], ref["tag_run_counts"])
        assert np.array_equal(res["pos_offsets"], ref["pos_offsets"])
        assert np.array_equal(res["positions"], ref["positions"])
        assert res["n_tag_overflow"] == ref["n_tag_overflow"]


def _fixture_reads(golden):
    d = os.path.join(golden, "bidirectional_test")
    reads = []
    for f in ("reads.txt", "test_reads.txt"):
        reads += [l for l in open(os.path.join(d, f)).read().split("\n") if l]
    return reads


@pytest.fixture(scope="module")
def xy(xy_paths, built):
    ri, tags = xy_paths
    return (P.Index(ri, tags), O.RIndex(ri), O.Tags(tags, O.TAGS_BYTECODE))


FORCE = {"dense": P.MODE_IMAGE_DENSE, "rl": P.MODE_IMAGE_RL, "dense2": P.MODE_IMAGE_DENSE2}
KIND = {"dense": P.IMAGE_DENSE, "rl": P.IMAGE_RL, "dense2": P.IMAGE_DENSE2}


@pytest.fixture(scope="module", params=["dense", "rl", "dense2"])
def xenc(x_index, request):
    """x index (encoded, no N) under every layout of the device rank image (dense, staged in LDS, is the automatic choice at this
    size; dense2 is the layout of everything that does not fit LDS and runs from global memory)"""
    ri, tags = x_index
    idx = P.Index(ri, tags, mode=P.MODE_COMPAT | FORCE[request.param])
    assert idx.info().image_kind == KIND[request.param]
    return (idx, O.RIndex(ri), O.Tags(tags, O.TAGS_COMPACT))


def test_device_present(built):
    assert P.device_count() >= 1
    assert "gfx950" in P.device_name(0)


@pytest.mark.parametrize("which", ["xy", "xenc"])
def test_rank_all_positions(which, xy, xenc):
    idx, ri, _ = xy if which == "xy" else xenc
    n = ri.n
    pos = np.concatenate([np.arange(0, n + 1, dtype=np.uint64), np.array([n + 1, n + 1000, 2**63], dtype=np.uint64)])
    got = idx.rank_batch(pos)
    true = idx.rank_batch(pos, true_codes=True)
    sm = ri.sym_map()
    present = [c for c, ch in enumerate(b"\nACGNT") if c == 0 or sm[ch] != 0]
    for i, p in enumerate(pos):
        exp = ri.rank_at_cached(int(min(p, n)))  # pos > n behaves like n (predecessor -> last block, totals)
        assert list(got[i][: ri.sigma]) == exp, (which, int(p))
        # header slots of absent codes carry the legacy block-cumulative quirk value in COMPAT images
        exp6 = ri.rank6_true(int(min(p, n)))
        assert [int(true[i][c]) for c in present] == [exp6[c] for c in present], (which, int(p))


@pytest.mark.parametrize("which", ["xy", "xenc"])
def test_extend_random(which, xy, xenc):
    idx, ri, _ = xy if which == "xy" else xenc
    rng = np.random.default_rng(7)
    n = ri.n
    m = 20000
    iv = np.zeros(m, dtype=P.BIINT_DTYPE)
    iv["forward"] = rng.integers(0, n, m)
    iv["size"] = 1 + (rng.random(m) * (n - iv["forward"])).astype(np.int64)
    iv["reverse"] = rng.integers(0, n, m)
    iv[:64]["forward"], iv[:64]["reverse"], iv[:64]["size"] = 0, 0, n
    alphabet = np.frombuffer(b"ACGTNacgtn\x00\n$XR", dtype=np.uint8)
    syms = alphabet[rng.integers(0, len(alphabet), m)]
    fw = rng.integers(0, 2, m).astype(np.uint8)
    got = idx.extend_batch(iv, syms, fw)
    for i in range(m):
        tri = (int(iv["forward"][i]), int(iv["reverse"][i]), int(iv["size"][i]))
        exp = (ri.fwd if fw[i] else ri.bwd)(tri, int(syms[i]))
        assert (int(got["forward"][i]), int(got["reverse"][i]), int(got["size"][i])) == exp, (i, tri, syms[i], fw[i])


@pytest.mark.parametrize("min_len,min_occ", SWEEP + [(3, 1), (0, 1), (1, 1), (12, 1), (13, 1), (5, 0), (5, 10**9)])
def test_fixture_reads_xy(xy, golden, min_len, min_occ):
    idx, ri, tags = xy
    cat, offs = O.pack_reads(_fixture_reads(golden))
    ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ)
    res = idx.find_mems(cat, offs, min_len, min_occ, tags=True)
    _assert_same(res, ref, True)


def test_known_answers_xy(xy, golden):
    """compat find_all_mems(read, 5, 1) on reads.txt with xy.ri"""
    idx, _, _ = xy
    reads = ["ACCCTAGAGTAT", "GGTAGCCATGCT", "TTTTGGAGGAGT", "CCCATAGTCGAA", "ATATATATATAT"]
    cat, offs = O.pack_reads(reads)
    res = idx.find_mems(cat, offs, 5, 1, tags=True)
    mems = [tuple(int(v) for v in m) for m in res["mems"]]
    assert list(res["mem_offsets"]) == [0, 0, 1, 3, 3, 3]
    assert mems == [(3, 10, 1381, 4), (4, 9, 5023, 12), (5, 11, 4399, 13)]
    po = res["pos_offsets"]
    assert list(res["positions"][po[0]:po[1]]) == [62467, 128008, 203779]
    assert list(res["positions"][po[1]:po[2]]) == [28697, 53276, 141325, 170009, 194588, 282637]
    assert list(res["positions"][po[2]:po[3]]) == [68623, 71684, 80923, 114688, 209935, 212996, 222235]
    assert list(res["tag_run_counts"]) == [3, 10, 10]


def test_edge_reads(xenc):
    idx, ri, tags = xenc
    reads = [b"A", b"", b"ACGT", b"N" * 30, b"acgtacgtacgtacgtacgtacgt", b"A" * 200, bytes([0, 1, 2, 65, 67, 71, 84] * 5),
             b"ACGTACGTAC" * 40, b"T" * 19, b"T" * 20, b"T" * 21, b"GATTACA\rGATTACA"]
    cat, offs = O.pack_reads(reads)
    for min_len, min_occ in [(0, 1), (1, 1), (4, 1), (20, 1), (20, 3), (500, 1)]:
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ)
        res = idx.find_mems(cat, offs, min_len, min_occ, tags=True)
        _assert_same(res, ref, True)


def test_empty_batch(xenc):
    idx, _, _ = xenc
    cat, offs = O.pack_reads([])
    res = idx.find_mems(cat, offs, 20, 1, tags=True)
    assert list(res["mem_offsets"]) == [0] and len(res["mems"]) == 0 and len(res["positions"]) == 0


@pytest.mark.parametrize("min_len,min_occ", SWEEP)
def test_synthetic_reads_x(xenc, golden, min_len, min_occ):
    idx, ri, tags = xenc
    seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
    cat, offs = W.sample_reads(seqs, 20000, 150, seed=42 + 1)
    ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
    res = idx.find_mems(cat, offs, min_len, min_occ, tags=True)
    _assert_same(res, ref, True)
    # on this no-N index every COMPAT search dies at the first T: min_len 20
    # finds nothing, shorter thresholds do
    assert (ref["mem_offsets"][-1] > 0) == (min_len < 20)


def test_synthetic_reads_xy_legacy(xy, golden):
    """real fixture pair: legacy .ri (quirk 1 on the block-cumulative slot) + ByteCode tags"""
    idx, ri, tags = xy
    seqs = W.load_sequences(os.path.join(golden, "bidirectional_test", "contigs_xy"))
    cat, offs = W.sample_reads(seqs, 20000, 150, seed=43)
    for min_len, min_occ in SWEEP:
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
        res = idx.find_mems(cat, offs, min_len, min_occ, tags=True)
        _assert_same(res, ref, True)


def test_strict_mode_matches_oracle_strict(x_index, xy_paths, golden):
    for (ri_path, tags_path), fmt, text in [(x_index, O.TAGS_COMPACT, "x.newline_separated"),
                                            (xy_paths, O.TAGS_BYTECODE, "bidirectional_test/contigs_xy")]:
        ri, tags = O.RIndex(ri_path), O.Tags(tags_path, fmt)
        seqs = W.load_sequences(os.path.join(golden, text))
        cat, offs = W.sample_reads(seqs, 5000, 100, seed=44)
        for force in (P.MODE_IMAGE_DENSE, P.MODE_IMAGE_RL, P.MODE_IMAGE_DENSE2):
            idx = P.Index(ri_path, tags_path, mode=P.MODE_STRICT | force)
            for min_len, min_occ in [(5, 1), (20, 1), (12, 2)]:
                ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, mode=O.MODE_STRICT, threads=O.lib().orc_max_threads())
                res = idx.find_mems(cat, offs, min_len, min_occ, tags=True)
                _assert_same(res, ref, True)
                assert ref["mem_offsets"][-1] > 0
            idx.close()


def test_tag_queries_all_sort_paths(xy):
    """run counts <= 64 (register bitonic), <= 2048 (LDS bitonic) and beyond (global bitonic)"""
    idx, ri, tags = xy
    rng = np.random.default_rng(11)
    n = ri.n
    st = rng.integers(0, n, 3000).astype(np.uint64)
    ln = np.concatenate([rng.integers(1, 60, 2000), rng.integers(60, 2500, 900), rng.integers(2500, n, 100)])
    en = np.minimum(st + ln.astype(np.uint64), np.uint64(n - 1))
    st[:3], en[:3] = [0, 0, 35], [n - 1, 0, 35]
    rn, po, pos, nover = idx.tag_query_batch(st, en)
    seen = set()
    for i in range(len(st)):
        ern, epos, eover = tags.query(int(st[i]), int(en[i]))
        assert int(rn[i]) == ern
        assert list(pos[po[i]:po[i + 1]]) == epos, i
        seen.add(0 if ern <= 64 else (1 if ern <= 2048 else 2))
    assert seen == {0, 1, 2}


@pytest.mark.parametrize("min_len", [10, 20])
def test_config2_one_million_reads(xenc, golden, min_len):
    """BASELINE configs[1]: x index, 1M synthetic 150-bp reads, bit-exact vs the CPU oracle.  min_len 10 is what bench.py's
    x workload runs (about 1.5 M MEMs and 1.7 M positions: emit path, slot compaction, CSR scans over > 2048 scan blocks and the
    whole tag pipeline at full size); at min_len 20 every COMPAT search on this no-N index dies (quirk 1): zero MEMs, only the
    extension counts and the empty CSR are compared there."""
    idx, ri, tags = xenc
    seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
    cat, offs = W.sample_reads(seqs, 1_000_000, 150, seed=42 + 2)
    ref = O.find_mems_batch(ri, tags, cat, offs, min_len, 1, threads=O.lib().orc_max_threads())
    res = idx.find_mems(cat, offs, min_len, 1, tags=True)
    _assert_same(res, ref, True)
    if min_len == 10:
        assert len(ref["mems"]) > 10**6 and ref["mem_offsets"][-1] == len(ref["mems"]) and len(ref["positions"]) > 10**6
    else:
        assert len(ref["mems"]) == 0 and ref["n_extensions"] > 10**7


def test_find_mems_function_per_start(xy, xenc, golden):
    """find_mems_function (algorithm.hpp:653-736) evaluated at EVERY start position of a set of reads (not only the starts the
    loop of find_all_mems visits): next start, pushed MEM and extension count vs the oracle"""
    for (idx, ri, _), text in ((xy, "bidirectional_test/contigs_xy"), (xenc, "x.newline_separated")):
        seqs = W.load_sequences(os.path.join(golden, text))
        cat, offs = W.sample_reads(seqs, 40, 90, seed=5)
        reads = [bytes(cat[int(offs[i]):int(offs[i + 1])]) for i in range(40)] + [b"", b"ACGT", b"N" * 12]
        cat, offs = O.pack_reads(reads)
        read_of = np.concatenate([np.full(len(r) + 1, i, dtype=np.uint64) for i, r in enumerate(reads)])
        xs = np.concatenate([np.arange(len(r) + 1, dtype=np.uint64) for r in reads])  # x = len included (only min_len 0 works there)
        for min_len, min_occ in [(5, 1), (0, 1), (12, 2), (200, 1)]:
            nx, mems, has, ne = idx.find_mems_function_batch(cat, offs, read_of, xs, min_len, min_occ)
            for q in range(len(xs)):
                enx, emem, ene = ri.find_mems_function(reads[int(read_of[q])], min_len, min_occ, int(xs[q]))
                assert int(nx[q]) == enx and int(ne[q]) == ene and bool(has[q]) == (emem is not None), (q, min_len)
                if emem is not None:
                    assert tuple(int(v) for v in mems[q]) == emem


def test_chunked_batches_equal_unchunked(xenc, xy, golden, monkeypatch):
    """batches whose worst-case MEM slots exceed the slot budget run in chunks of consecutive reads"""
    for (idx, ri, tags), text, params in ((xenc, "x.newline_separated", (5, 1)), (xy, "bidirectional_test/contigs_xy", (10, 1))):
        seqs = W.load_sequences(os.path.join(golden, text))
        cat, offs = W.sample_reads(seqs, 30000, 150, seed=123)
        whole = idx.find_mems(cat, offs, *params, tags=True)
        monkeypatch.setenv("PGX_SLOT_BUDGET_MB", "1")  # 32768 slots -> ~130 chunks
        b = idx.batch(cat, offs)
        b.run(params[0], params[1], P.RUN_TAGS | P.RUN_TIMING)
        chunked, launches = b.result(), b.timing().find_mems_launches
        b.free()
        monkeypatch.delenv("PGX_SLOT_BUDGET_MB")
        assert launches > 50
        _assert_same(chunked, whole, True)
        ref = O.find_mems_batch(ri, tags, cat, offs, *params, threads=O.lib().orc_max_threads())
        _assert_same(chunked, ref, True)


def test_sharded_equals_unsharded(xenc, golden):
    """contiguous read slices processed independently and concatenated in rank order are
    bit-identical to the unsharded batch (what bench.py --gpus N relies on)."""
    import pgx_shard as S

    idx, ri, tags = xenc
    seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
    cat, offs = W.sample_reads(seqs, 30001, 150, seed=99)
    whole = idx.find_mems(cat, offs, 10, 1, tags=True)
    for world in (2, 3, 8):
        parts = []
        for rank in range(world):
            c, o = S.shard_reads(cat, offs, rank, world)
            parts.append(idx.find_mems(c, o, 10, 1, tags=True))
        merged = S.merge_results(parts)
        _assert_same(merged, whole, True)
    ref = O.find_mems_batch(ri, tags, cat, offs, 10, 1, threads=O.lib().orc_max_threads())
    _assert_same(whole, ref, True)


def test_sharded_through_the_c_abi(xenc, golden):
    """pgx_find_mems_sharded: the slices run on host threads of the library, one batch and stream each (here every slice on device 0:
    the code path of an N-GPU node); concatenated in slice order = the unsharded result"""
    import pgx_shard as S

    idx, ri, tags = xenc
    seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
    cat, offs = W.sample_reads(seqs, 40003, 150, seed=101)
    ref = O.find_mems_batch(ri, tags, cat, offs, 10, 1, threads=O.lib().orc_max_threads())
    for devices in ([0], [0, 0], [0, 0, 0, 0, 0]):
        parts, first = P.find_mems_sharded(idx, devices, cat, offs, 10, 1, tags=True)
        assert first[0] == 0 and first[-1] == 40003 and len(parts) == len(devices)
        _assert_same(S.merge_results(parts), ref, True)
    with pytest.raises(P.PgxError):
        P.find_mems_sharded(idx, [0, 99], cat, offs, 10, 1)


def test_sigma6_pangenome_both_images(workdir):
    """sigma = 6 synthetic pangenome (N runs, both strands): COMPAT == STRICT there, and both image layouts must agree
    with the oracle on reads that include N runs, sequence ends and reverse complements"""
    text = os.path.join(workdir, "p6.txt")
    W.synth_pangenome_text(text, base_len=60000, n_hap=4, seed=11, n_runs=3, n_run_len=(100, 2000))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "p6")[:2]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    assert ri.sigma == 6 and ri.has_N
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 20000, 150, seed=61)
    for min_len, min_occ in [(20, 1), (12, 4)]:
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
        strict = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, mode=O.MODE_STRICT, threads=O.lib().orc_max_threads())
        # same MEMs; the extension counts differ (STRICT rejects pattern[len] = 0 at once, COMPAT walks on from the endmarker)
        assert ref["mems"].tobytes() == strict["mems"].tobytes() and ref["n_extensions"] != strict["n_extensions"]
        for mode in (P.MODE_COMPAT | P.MODE_IMAGE_DENSE, P.MODE_COMPAT | P.MODE_IMAGE_RL, P.MODE_STRICT | P.MODE_IMAGE_DENSE,
                     P.MODE_STRICT | P.MODE_IMAGE_RL, P.MODE_COMPAT | P.MODE_IMAGE_DENSE2, P.MODE_STRICT | P.MODE_IMAGE_DENSE2):
            idx = P.Index(ri_path, tags_path, mode=mode)
            _assert_same(idx.find_mems(cat, offs, min_len, min_occ, tags=True), strict if mode & 1 else ref, True)
            idx.close()


def test_long_and_ragged_reads(workdir):
    """reads from 1 to 12 000 bp in one batch (beyond the heavy-read length limit, many slots per read), both layouts"""
    text = os.path.join(workdir, "long_reads.txt")
    W.synth_pangenome_text(text, base_len=50000, n_hap=3, seed=21, n_runs=2, n_run_len=(100, 1500))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "long_reads")[:2]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(text)
    rng = np.random.default_rng(8)
    reads = []
    for ln in [1, 2, 19, 20, 21, 150, 151, 1000, 4095, 4096, 4097, 12000] + [int(v) for v in rng.integers(1, 6000, 120)]:
        s = seqs[int(rng.integers(0, len(seqs)))]
        a = int(rng.integers(0, len(s) - ln))
        r = np.array(s[a:a + ln])
        flip = rng.random(ln) < 0.01
        r[flip] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(flip.sum()))]
        reads.append(bytes(r))
    reads += [bytes(seqs[1][-2048:]), bytes(seqs[0][-700:])]  # sequence ends: quadratic reads for the heavy-read kernel
    cat, offs = O.pack_reads(reads)
    for min_len, min_occ in [(20, 1), (300, 2)]:
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
        for force in (P.MODE_IMAGE_DENSE, P.MODE_IMAGE_RL, P.MODE_IMAGE_DENSE2):
            idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | force)
            _assert_same(idx.find_mems(cat, offs, min_len, min_occ, tags=True), ref, True)
            idx.close()


def test_batch_reuse_and_parameter_changes(xenc, golden):
    """one long-lived batch: new reads (fewer, then more), then the same reads under other (min_len, min_occ): cached
    slot offsets and chunk plans must follow"""
    idx, ri, tags = xenc
    seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
    sets = [W.sample_reads(seqs, n, ln, seed=70 + i) for i, (n, ln) in enumerate([(5000, 150), (100, 60), (20000, 150), (3, 150)])]
    b = idx.batch(*sets[0])
    try:
        for i, (cat, offs) in enumerate(sets):
            if i:
                b.upload(cat, offs)
            for min_len, min_occ in [(10, 1), (5, 1), (10, 1), (12, 3)]:
                b.run(min_len, min_occ, P.RUN_TAGS)
                _assert_same(b.result(), O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads()), True)
    finally:
        b.free()


def test_concurrent_batches_from_host_threads(x_index, golden):
    """the index handle is shared by host threads, each with its own batch (SURVEY 8b threading row)"""
    import threading
    ri_path, tags_path = x_index
    idx = P.Index(ri_path, tags_path)  # device image created by whichever thread comes first
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
    work = [W.sample_reads(seqs, 4000 + 500 * t, 150, seed=90 + t) for t in range(4)]
    refs = [O.find_mems_batch(ri, tags, c, o, 10, 1, threads=4) for c, o in work]
    errors = []

    def run(t):
        try:
            cat, offs = work[t]
            b = idx.batch(cat, offs)
            for _ in range(5):
                b.run(10, 1, P.RUN_TAGS)
                _assert_same(b.result(), refs[t], True)
            b.free()
        except Exception as e:  # noqa: BLE001 -- reported below with the thread id
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("layout", ["dense", "dense2"])
@pytest.mark.parametrize("env", [{"PGX_FM_NARROW": "0"}, {"PGX_FM_NARROW_FORCE_REDO": "1"}])
def test_wide_state_and_forced_redo(x_index, golden, monkeypatch, env, layout):
    """dense images of short BWTs run with 32-bit interval state; the 64-bit kernels (PGX_FM_NARROW=0) and the repeat of a
    chunk after a (here: simulated) 32-bit overflow must give the same answers"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ri_path, tags_path = x_index
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | FORCE[layout])
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
    cat, offs = W.sample_reads(seqs, 20000, 150, seed=97)
    monkeypatch.setenv("PGX_SLOT_BUDGET_MB", "20")  # several chunks: the repeat happens per chunk
    for min_len, min_occ in [(10, 1), (5, 2)]:
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
        b = idx.batch(cat, offs)
        b.run(min_len, min_occ, P.RUN_TAGS | P.RUN_TIMING)
        _assert_same(b.result(), ref, True)
        launches = b.timing().find_mems_launches
        b.free()
        assert launches >= 3


def _packed_arrays(cat, offs, pinned):
    alloc = P.pinned_array if pinned else (lambda n, dt: np.zeros(n, dtype=dt))
    packed = alloc(max((len(cat) + 15) // 16, 1), np.uint32)
    ids, side = alloc(len(offs), np.uint64), alloc(len(cat) + 16, np.uint8)
    ns, _ = P.pack_reads(cat, offs, packed, ids, side)
    return packed, ids, side, ns


def test_packed_upload_equals_byte_upload(workdir, x_index):
    """pgx_batch_upload_packed (the reads packed to two bits per symbol by the caller, reads with a byte outside A C G T listed with their bytes;
    what the find_mems CLI uploads) gives the bytes pgx_batch_upload gives: on a sigma = 6 pangenome under the automatic layout (two-step kernel
    over the packed words, listed reads on the side stream) and forced dense2 (every kernel reads the bytes rebuilt on the device), and on the x
    index (image in LDS); pinned and ordinary host memory; a long-lived batch refilled in both forms alternately; reads with N, lower case, \\0,
    empty reads, lengths that are no multiple of 16.  Reference unit: the per-read loop src/find_mems.cpp:94-139."""
    text = os.path.join(workdir, "pk6.txt")
    W.synth_pangenome_text(text, base_len=300_000, n_hap=4, seed=12, n_runs=3, n_run_len=(100, 2000))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "pk6")[:2]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 40_000, 150, seed=62, n_frac=0.02)
    rng = np.random.default_rng(9)
    extra = [b"", b"N" * 150, b"acgt" * 30, b"ACGT\x00ACGTACGTACGTACGTACGTACGT", bytes(seqs[0][-150:]), bytes(seqs[1][:150]), b"A", b"", bytes(seqs[2][500:533]),
             bytes(seqs[3][100:351]), b"ACGTN"]
    for _ in range(300):
        s = seqs[int(rng.integers(0, len(seqs)))]
        ln = int(rng.integers(1, 330))
        a = int(rng.integers(0, len(s) - ln))
        extra.append(bytes(s[a:a + ln]))
    ecat, eoffs = O.pack_reads(extra)
    cat = np.concatenate([cat, ecat]); offs = np.concatenate([offs, eoffs[1:] + offs[-1]])
    cat2, offs2 = W.sample_reads(seqs, 30_000, 150, seed=63)  # a second batch without a single listed read ...
    keep = ~(cat2.reshape(-1, 150) == ord("N")).any(axis=1)
    cat2 = cat2.reshape(-1, 150)[keep].reshape(-1); offs2 = np.arange(int(keep.sum()) + 1, dtype=np.uint64) * np.uint64(150)
    for mode in (P.MODE_COMPAT, P.MODE_COMPAT | P.MODE_IMAGE_DENSE2):
        idx = P.Index(ri_path, tags_path, mode=mode)
        assert idx.info().image_pairs == (1 if mode == P.MODE_COMPAT else 0)
        for min_len, min_occ in ((20, 1), (12, 2)):
            ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
            ref2 = O.find_mems_batch(ri, tags, cat2, offs2, min_len, min_occ, threads=O.lib().orc_max_threads())
            b = idx.batch_empty()
            for pinned in (True, False):
                packed, ids, side, ns = _packed_arrays(cat, offs, pinned)
                assert ns > 700  # reads that overlap an N run, lower case, \0
                b.upload_packed(packed, offs, ids, side, ns)
                b.run(min_len, min_occ, P.RUN_TAGS | P.RUN_TIMING)
                t1 = b.timing()
                res = b.result()
                _assert_same(res, ref, True)
                assert res["n_extensions"] == ref["n_extensions"]
                b.run(min_len, min_occ, P.RUN_TAGS | P.RUN_TIMING)  # the same reads again: nothing left to prepare
                assert b.timing().ms_per_upload == 0.0 and t1.ms_per_upload > 0.0
                _assert_same(b.result(), ref, True)
                if mode == P.MODE_COMPAT:
                    assert t1.pairs_reads == (4 if min_occ <= 1 else 2)  # the two-step kernel over the packed words as they came from the host (4: + narrow forward stages through the text, min_occ <= 1)
                b.upload(cat2, offs2)  # refilled as bytes ...
                b.run(min_len, min_occ, P.RUN_TAGS | P.RUN_TIMING)
                assert b.timing().ms_per_upload > 0.0
                _assert_same(b.result(), ref2, True)
                p2, i2, s2, n2 = _packed_arrays(cat2, offs2, pinned)  # ... and packed, with an empty side list
                assert n2 == 0
                b.upload_packed(p2, offs2, i2, s2, 0)
                b.run(min_len, min_occ, P.RUN_TAGS)
                r2 = b.result()
                _assert_same(r2, ref2, True)
                assert r2["n_extensions"] == ref2["n_extensions"]
            b.free()
        idx.close()
    # the x index (no N in the index, image in LDS, COMPAT quirks): the kernels read the rebuilt bytes
    xri, xtags = x_index
    xr, xt = O.RIndex(xri), O.Tags(xtags, O.TAGS_COMPACT)
    xref = O.find_mems_batch(xr, xt, cat, offs, 10, 1, threads=O.lib().orc_max_threads())
    idx = P.Index(xri, xtags, mode=P.MODE_COMPAT)
    b = idx.batch_empty()
    packed, ids, side, ns = _packed_arrays(cat, offs, True)
    b.upload_packed(packed, offs, ids, side, ns)
    b.run(10, 1, P.RUN_TAGS)
    _assert_same(b.result(), xref, True)
    # refused uploads leave the batch usable: offsets that do not start at 0, listed ids out of order
    with pytest.raises(P.PgxError):
        b.upload_packed(packed, offs + np.uint64(16), ids, side, ns)
    bad_ids = ids.copy(); bad_ids[:2] = bad_ids[:2][::-1]
    with pytest.raises(P.PgxError):
        b.upload_packed(packed, offs, bad_ids, side, ns)
    b.run(10, 1, P.RUN_TAGS)
    _assert_same(b.result(), xref, True)
    b.free()
    idx.close()

"""The two-step PAIRS kernel (pgx_find_mems_pairs_kernel: two extensions per cache line; pgx_image.h): MEMs, tag positions and
the exact extension count equal the oracle's, whatever the seed depth, the mode, min_len (odd, even, below the seed depth) and
min_occ; reads that meet endmarkers or N in the BWT are handed to the dense2 kernel and come out the same."""
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
    b.run(min_len, min_occ, flags=P.RUN_TAGS | P.RUN_TIMING)
    res, t = b.result(), b.timing()
    stats = (1 if t.pairs_reads else 0, int(t.pairs_other_steps))  # (pairs_reads also says which variant ran: 1 byte windows, 2 packed reads, 3 cooperative fetches)
    b.free()
    return res, stats


@pytest.fixture(scope="module")
def pan(workdir):
    text = os.path.join(workdir, "pairpan.txt")
    W.synth_pangenome_text(text, base_len=150000, n_hap=4, seed=33, n_runs=3, n_run_len=(100, 3000))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "pairpan")[:2]
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 30000, 150, seed=78)
    rng = np.random.default_rng(4)
    extra = []
    for _ in range(600):  # odd bytes, short reads, reads of every length parity
        s = seqs[int(rng.integers(0, len(seqs)))]
        ln = int(rng.integers(1, 220))
        a = int(rng.integers(0, len(s) - ln))
        r = bytearray(bytes(s[a:a + ln]))
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, ln))] = int(rng.choice(np.frombuffer(b"Nacgt\x00$", dtype=np.uint8)))
        extra.append(bytes(r))
    for s in seqs[:4]:  # reads that end a sequence (heavy) and reads that start one
        extra.append(bytes(s[-150:]))
        extra.append(bytes(s[:150]))
    ecat, eoffs = O.pack_reads(extra)
    cat = np.concatenate([cat, ecat])
    offs = np.concatenate([offs, eoffs[1:] + offs[-1]])
    return ri_path, tags_path, cat, offs


_REF = {}  # the oracle's answers, computed once per (mode, min_len, min_occ): they do not depend on the device layout under test


def _oracle(ri, tags, cat, offs, min_len, min_occ, omode):
    key = (omode, min_len, min_occ, len(offs))
    if key not in _REF:
        _REF[key] = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, mode=omode, threads=O.lib().orc_max_threads())
    return _REF[key]


@pytest.mark.parametrize("seed_k,psyms", [("0", None), ("7", "96"), ("7", "64"), (None, "96"), (None, "64")])
def test_pairs_kernel_equals_the_oracle(pan, monkeypatch, seed_k, psyms):
    """psyms: the stride of the PAIRS image (pgx_image.h: blocks of 96 positions every 96 positions, or every 64)"""
    ri_path, tags_path, cat, offs = pan
    if psyms:
        monkeypatch.setenv("PGX_PAIRS_STRIDE", psyms)
    if seed_k is None:
        monkeypatch.delenv("PGX_SEED_K", raising=False)
    else:
        monkeypatch.setenv("PGX_SEED_K", seed_k)
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    n_reads = len(offs) - 1
    for mode, omode in ((P.MODE_COMPAT, O.MODE_COMPAT), (P.MODE_STRICT, O.MODE_STRICT)):
        idx = P.Index(ri_path, tags_path, mode=mode | P.MODE_IMAGE_PAIRS)
        assert idx.info().image_pairs == 1 and idx.info().image_kind == P.IMAGE_DENSE2 and idx.info().pairs_stride == int(psyms or 64)
        idx_seed_k = int(seed_k) if seed_k is not None else 10  # automatic: depth 12 for n = 1.2 M and the second table of depth 10
        for min_len, min_occ in [(20, 1), (21, 1), (8, 1), (7, 1), (3, 1), (1, 1), (0, 1), (20, 2), (25, 9), (20, 0), (40, 1), (33, 3)]:
            ref = _oracle(ri, tags, cat, offs, min_len, min_occ, omode)
            res, (used, redo) = _run(idx, cat, offs, min_len, min_occ)
            _same(res, ref)
            seeded = seed_k != "0" and min_len >= idx_seed_k
            assert used == (1 if seeded else 0)  # the kernel runs behind the seed table only
            if seeded and seed_k is None and min_occ <= 1:  # (a larger min_occ lets fewer seeds apply: unseeded stages start wide and are handed on)
                assert redo < 2 * n_reads, (min_len, redo)  # extensions taken through the dense2 image: the two-step path does the work (of ~200 extensions per read)
        idx.close()


def test_pairs_kernel_heavy_reads_and_switch(pan, monkeypatch):
    ri_path, tags_path, cat, offs = pan
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    ref = O.find_mems_batch(ri, tags, cat, offs, 20, 1, threads=O.lib().orc_max_threads())
    for env in ({"PGX_FM_HEAVY_EXT": "40"}, {"PGX_FM_PAIRS": "0"}, {"PGX_FM_HEAVY_EXT": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
        res, (used, _) = _run(idx, cat, offs, 20, 1)
        _same(res, ref)
        assert used == (0 if "PGX_FM_PAIRS" in env else 1)
        idx.close()
        for k in env:
            monkeypatch.delenv(k)


def test_pairs_kernel_on_an_index_full_of_special_runs(workdir, golden, monkeypatch):
    """a small multi-sequence index: most blocks hold an endmarker or N neighbour, most reads take the fallback -- same results"""
    ri_path, tags_path = W.build_index_from_rlbwt(os.path.join(golden, "med_test.rl_bwt"), workdir, "pairs_med")
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    rng = np.random.default_rng(8)
    reads = ["".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(1, 60)))) for _ in range(3000)]
    cat, offs = O.pack_reads([r.encode() for r in reads])
    monkeypatch.setenv("PGX_SEED_K", "6")
    idx = P.Index(ri_path, tags_path, mode=P.MODE_STRICT | P.MODE_IMAGE_PAIRS)
    for min_len, min_occ in [(3, 1), (6, 1), (7, 2), (12, 1)]:
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, mode=O.MODE_STRICT, threads=O.lib().orc_max_threads())
        res, (used, _) = _run(idx, cat, offs, min_len, min_occ)
        _same(res, ref)
        assert used == (1 if min_len >= 6 else 0)
    idx.close()


def test_pairs_kernel_in_chunks_and_after_a_forced_repeat(pan, monkeypatch):
    """batches processed in chunks of the slot budget (the per-chunk counters, cursor and hand-on list are reset per chunk) and the
    repeat-in-64-bits path of the dense2 kernel that serves the handed-on reads"""
    ri_path, tags_path, cat, offs = pan
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    ref = O.find_mems_batch(ri, tags, cat, offs, 20, 1, threads=O.lib().orc_max_threads())
    for env in ({"PGX_SLOT_BUDGET_MB": "8"}, {"PGX_FM_NARROW_FORCE_REDO": "1"}, {"PGX_SLOT_BUDGET_MB": "8", "PGX_FM_NARROW_FORCE_REDO": "1", "PGX_SPEC": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
        b = idx.batch(cat, offs)
        for _ in range(2):  # the second run of the same batch is sized speculatively (unless switched off)
            b.run(20, 1, flags=P.RUN_TAGS | P.RUN_TIMING)
            _same(b.result(), ref)
            assert b.timing().pairs_reads != 0  # (1: byte windows -- chunked batches; 2: reads packed in LDS; 3: cooperative fetches)
        if "PGX_SLOT_BUDGET_MB" in env:
            assert b.timing().find_mems_launches > 1
        b.free()
        idx.close()
        for k in env:
            monkeypatch.delenv(k)


def test_pairs_kernel_empty_and_tiny_batches(pan):
    ri_path, tags_path, cat, offs = pan
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
    for n in (0, 1, 3, 65):
        c, o = cat[:int(offs[n])], offs[:n + 1]
        ref = O.find_mems_batch(ri, tags, c, o, 20, 1)
        _same(idx.find_mems(c, o, 20, 1, tags=True), ref)
    idx.close()


@pytest.mark.parametrize("read_len,variant", [(250, 2), (353, 2), (354, 1), (600, 1), (1000, 1)])
def test_pairs_kernel_read_lengths(pan, read_len, variant):
    """read lengths around the limit of the packed-reads form (24 words of LDS per lane with one of padding: up to 353 symbols at the worst phase) and beyond it, where the
    kernel reads 16-byte windows of the read bytes instead; some reads with N, reads that end a sequence"""
    ri_path, tags_path, _, _ = pan
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(os.path.join(os.path.dirname(ri_path), "pairpan.txt"))
    cat, offs = W.sample_reads(seqs, 3000, read_len, seed=1000 + read_len)
    extra = [bytes(s[-read_len:]) for s in seqs[:2]] + [bytes(s[:read_len]) for s in seqs[:2]] + [b"ACGT" * (read_len // 4), bytes(seqs[0][:read_len - 1])]
    ecat, eoffs = O.pack_reads(extra)
    cat = np.concatenate([cat, ecat]); offs = np.concatenate([offs, eoffs[1:] + offs[-1]])
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
    for min_len, min_occ in [(20, 1), (31, 2)]:
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
        b = idx.batch(cat, offs)
        b.run(min_len, min_occ, flags=P.RUN_TAGS | P.RUN_TIMING)
        _same(b.result(), ref)
        assert b.timing().pairs_reads == (4 if (variant == 2 and min_occ <= 1) else variant)  # (4: packed reads + narrow forward stages through the text, min_occ <= 1)
        b.free()
    idx.close()


def test_pairs_kernel_with_mostly_lower_case_reads(pan):
    """more 16-byte chunks with a byte outside ACGT than the classification lists: no second-stream launch, the pairs kernel hands
    such reads on itself (no seed applies to them) -- same results"""
    ri_path, tags_path, cat, offs = pan
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    n = 3000
    c = cat[:int(offs[n])].copy()
    o = offs[:n + 1]
    lower = np.arange(len(c)) % 3 != 0  # two bytes of three in lower case
    c[lower] = np.where((c[lower] >= 65) & (c[lower] <= 90), c[lower] + 32, c[lower])
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
    for min_len in (20, 12):
        ref = O.find_mems_batch(ri, tags, c, o, min_len, 1, threads=O.lib().orc_max_threads())
        res, (used, _) = _run(idx, c, o, min_len, 1)
        _same(res, ref)
        assert used == 1
    idx.close()


def _random_index_case(workdir, monkeypatch, seed, force):
    rng = np.random.default_rng(1000 + seed)
    seqs = []
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(800, 6000))))
    for h in range(int(rng.integers(2, 7))):
        s = list(base)
        for i in rng.integers(0, len(s), max(1, len(s) // 60)):
            s[i] = "ACGT"[rng.integers(0, 4)]
        for _ in range(int(rng.integers(0, 4))):
            ln = int(rng.integers(1, 80))
            a = int(rng.choice([0, len(s) - ln, int(rng.integers(0, len(s) - ln))]))
            s[a:a + ln] = "N" * ln
        s = "".join(s)
        seqs.append(s)
        if seed % 2:
            seqs.append(s[::-1].translate(str.maketrans("ACGTN", "TGCAN")))
    text = os.path.join(workdir, "pairs_rand_%d.txt" % seed)
    with open(text, "w") as f:
        for s in seqs:
            f.write(s + "\n")
    ri_path, tags_path = W.build_index_from_text(text, workdir, "pairs_rand_%d" % seed)[:2]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    reads = []
    for _ in range(1500):
        s = seqs[int(rng.integers(0, len(seqs)))]
        ln = int(rng.integers(1, 120))
        a = int(rng.choice([0, max(0, len(s) - ln), int(rng.integers(0, max(1, len(s) - ln)))]))
        r = bytearray(s[a:a + ln].encode())
        for _ in range(int(rng.integers(0, 3))):
            if r:
                r[int(rng.integers(0, len(r)))] = int(rng.choice(np.frombuffer(b"ACGTNa", dtype=np.uint8)))
        reads.append(bytes(r))
    cat, offs = O.pack_reads(reads)
    if force:
        monkeypatch.setenv("PGX_SEED_K", str(int(rng.integers(3, 7))))
        monkeypatch.setenv("PGX_PAIRS_STRIDE", "96" if seed % 2 == 0 else "64")  # both strides of the PAIRS image
    else:
        monkeypatch.delenv("PGX_SEED_K", raising=False)
    for mode, omode in ((P.MODE_COMPAT, O.MODE_COMPAT), (P.MODE_STRICT, O.MODE_STRICT)):
        try:
            idx = P.Index(ri_path, tags_path, mode=mode | force)
        except P.PgxError as e:  # (a text without N in COMPAT: the quirk tables do not qualify)
            assert force and e.code == P.ERR_UNSUPPORTED and mode == P.MODE_COMPAT and not ri.has_N
            continue
        if force:
            assert idx.info().pairs_stride == (96 if seed % 2 == 0 else 64)
        for min_len, min_occ in [(7, 1), (8, 1), (12, 2), (20, 1), (10, 1)]:
            ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, mode=omode, threads=O.lib().orc_max_threads())
            res, (used, _) = _run(idx, cat, offs, min_len, min_occ)
            _same(res, ref)
            if force:
                assert used == 1
        idx.close()


@pytest.mark.parametrize("seed", list(range(10)))
def test_pairs_kernel_on_random_small_indexes(workdir, monkeypatch, seed):
    """random texts (2-6 sequences, N runs anywhere incl. at sequence starts and ends, some in both strands): so small that a large share of
    the blocks is flagged -- reads are handed on in the middle of a search and resumed from their current start position all the time"""
    _random_index_case(workdir, monkeypatch, seed, P.MODE_IMAGE_PAIRS)


@pytest.mark.parametrize("seed", list(range(10, 18)))
def test_automatic_layout_on_random_small_indexes(workdir, monkeypatch, seed):
    """the same texts under the automatic layout: the image staged in LDS with its seed table of depth 10 and the end table (the smaller
    ones), the 64-byte image in global memory + the pairs image (the larger ones)"""
    _random_index_case(workdir, monkeypatch, seed, 0)


@pytest.mark.parametrize("haps", [24, 48, 80])
def test_wide_intervals_and_the_run_continuation(workdir, monkeypatch, haps):
    """Pangenomes of many haplotypes (the reference's README pipeline targets ~90: /root/reference/README.md:76-100): intervals are ~#haplotypes
    positions wide, wider than the 32 a block every 64 positions always covers.  The PAIRS block says how far the pair / the first symbol of its last
    position goes on behind it (pgx_image.h "run continuation"): an interval that ends inside that stretch is answered from ONE line.  Same bytes as the
    oracle with and without the fields, narrow / 64-bit / cooperative fetches, both strides; and fewer lines with them."""
    text = os.path.join(workdir, "wide_iv_%d.txt" % haps)
    W.synth_pangenome_text(text, base_len=3_000_000 // haps, n_hap=haps, seed=50 + haps, n_runs=2, n_run_len=(100, 1500))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "wide_iv_%d" % haps)[:2]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 30_000, 150, seed=9)
    extra = [bytes(seqs[0][-150:]), bytes(seqs[1][:150]), b"ACGT" * 30, b"", b"N" * 20]
    ecat, eoffs = O.pack_reads(extra)
    cat = np.concatenate([cat, ecat]); offs = np.concatenate([offs, eoffs[1:] + offs[-1]])
    monkeypatch.setenv("PGX_SB_SHIFT", "6")
    refs = {pr: O.find_mems_batch(ri, tags, cat, offs, pr[0], pr[1], threads=O.lib().orc_max_threads()) for pr in ((20, 1), (13, 5))}
    lines = {}
    for ext, stride, wide, coop in (("0", "64", 0, None), (None, "64", 0, None), (None, "96", 0, None), ("0", "64", 1, None), (None, "64", 1, None), (None, "64", 0, "1"), (None, "96", 1, "1")):
        for k, v in (("PGX_PAIRS_EXT", ext), ("PGX_PAIRS_STRIDE", stride), ("PGX_FM_COOP", coop)):
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS | (P.MODE_IMAGE_WIDE if wide else 0))
        pv = idx.image_view(20).view(np.uint32).reshape(-1, 32)
        assert bool((pv[:, 17] >> 24).any()) == (ext is None)
        for pr, ref in refs.items():
            b = idx.batch(cat, offs)
            b.run(pr[0], pr[1], P.RUN_TAGS | P.RUN_TIMING)
            t = b.timing()
            assert t.pairs_reads == (3 if coop else (4 if (not wide and pr[1] <= 1) else 2))  # (4: packed reads + forward stages through the text: narrow images, min_occ <= 1)
            _same(b.result(), ref)
            if pr == (20, 1):
                lines[(ext, stride, wide, coop)] = int(t.main_lines)
            b.free()
        idx.close()
    with_ext, without = lines[(None, "64", 1, None)], lines[("0", "64", 1, None)]  # (the 64-bit variant: the narrow one takes these intervals through the text)
    assert lines[(None, "64", 1, None)] == lines[(None, "64", 0, "1")]  # the same trips in the 64-bit and the cooperative variant
    assert lines[(None, "64", 0, None)] < (0.5 if haps < 80 else 0.75) * with_ext, (haps, lines)  # intervals of up to 128 occurrences through the text and the table of common prefixes
    # (24 haplotypes: intervals stay below the 32 positions a block every 64 always covers -- nothing to gain, nothing lost; 48 and 80: second lines saved)
    assert with_ext <= without and (haps == 24 or with_ext < 0.9 * without), (haps, with_ext, without)


@pytest.mark.parametrize("haps,stride", [(4, "64"), (12, "64"), (12, "96"), (40, "64")])
def test_forward_stages_through_the_text(workdir, monkeypatch, haps, stride):
    """pgx_find_mems_pairs_kernel<.., LCE>: where a forward stage's interval is narrow (<= 128 occurrences; <= 16 without the table of common prefixes) the kernel
    finishes it from the suffix array and the text -- longest match over the occurrences and the occurrences that reach it -- instead of two symbols per trip (algorithm.hpp:676-700 is the
    loop it replaces: the forward extensions of find_mems_function).  Same bytes as the oracle and as the stepwise kernel (PGX_FM_LCE=0), extension counts
    included, for min_occ 0 / 1 (min_occ > 1 never takes this path); reads that end / start a sequence, reads over N runs (flagged text lines), short reads,
    reads longer than the 144 symbols a text window holds; 40 haplotypes: intervals wider than sixteen take the path only with the table; and fewer lines where it is taken.
    With img.lce_lcp (the default; PGX_FM_LCP=0 without) the occurrences after a compared one follow from the common prefixes of neighbouring suffixes
    (pgx_lce_lcp_kernel; tests/test_lce_math.py has the arithmetic): same bytes again, fewer lines again."""
    text = os.path.join(workdir, "lce_%d.txt" % haps)
    W.synth_pangenome_text(text, base_len=1_200_000 // haps, n_hap=haps, seed=70 + haps, n_runs=3, n_run_len=(60, 900))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "lce_%d" % haps)[:2]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 25_000, 150, seed=11, n_frac=0.03)
    rng = np.random.default_rng(5)
    extra = [bytes(seqs[0][-150:]), bytes(seqs[1][:150]), bytes(seqs[-1][-150:]), b"ACGT" * 30, b"", b"A", b"N" * 30]
    for _ in range(400):  # ragged lengths, some longer than a text window
        sq = seqs[int(rng.integers(0, len(seqs)))]
        ln = int(rng.integers(1, 340))
        a = int(rng.integers(0, len(sq) - ln))
        extra.append(bytes(sq[a:a + ln]))
    ecat, eoffs = O.pack_reads(extra)
    cat = np.concatenate([cat, ecat]); offs = np.concatenate([offs, eoffs[1:] + offs[-1]])
    monkeypatch.setenv("PGX_PAIRS_STRIDE", stride)
    refs = {pr: O.find_mems_batch(ri, tags, cat, offs, pr[0], pr[1], threads=O.lib().orc_max_threads()) for pr in ((20, 1), (12, 1), (25, 0), (20, 3))}
    lines = {}
    for lce, lcp in (("0", None), (None, "0"), (None, None)):  # stepwise; every occurrence compared with the text; the common-prefix table (the default)
        for name, v in (("PGX_FM_LCE", lce), ("PGX_FM_LCP", lcp)):
            if v is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, v)
        idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
        for pr, ref in refs.items():
            b = idx.batch(cat, offs)
            for _ in range(2):
                b.run(pr[0], pr[1], P.RUN_TAGS | P.RUN_TIMING)
                t = b.timing()
                assert t.pairs_reads == (4 if (lce is None and pr[1] <= 1) else 2), (lce, pr, t.pairs_reads)
                _same(b.result(), ref)
            if pr == (20, 1):
                lines[(lce, lcp)] = int(t.main_lines)
            b.free()
        idx.close()
    if haps <= 12:
        assert lines[(None, "0")] < (0.8 if haps <= 4 else 0.95) * lines[("0", None)], (haps, lines)  # (without the table: a text line AND the line of its suffix array entry per occurrence)
        assert lines[(None, None)] < (0.9 if haps >= 4 else 1.02) * lines[(None, "0")], (haps, lines)
    else:  # (intervals of ~40 occurrences: without the table the path is not taken, with it most forward stages are)
        assert lines[(None, "0")] <= 1.02 * lines[("0", None)] and lines[(None, None)] < 0.6 * lines[("0", None)], (haps, lines)


def test_no_text_comparison_on_a_forward_only_collection(workdir, monkeypatch):
    """the FMD index answers forward extensions through the reverse complement (src/r-index.cpp:758-764): on a collection that does not hold its sequences in
    both orientations that is NOT "the occurrences of the longer pattern", so the text comparison must not stand in for it -- the LCE image is not built and
    the kernel runs stepwise (the reference's arithmetic, whatever it means there), same bytes as the oracle"""
    rng = np.random.default_rng(77)
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 20000))
    text = os.path.join(workdir, "fwd_only.txt")
    with open(text, "w") as f:
        for h in range(4):
            s = list(base)
            for i in rng.integers(0, len(s), 200):
                s[i] = "ACGT"[rng.integers(0, 4)]
            if h == 1:
                s[5000:5040] = "N" * 40
            f.write("".join(s) + "\n")
    ri_path, tags_path = W.build_index_from_text(text, workdir, "fwd_only")[:2]
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_COMPACT)
    seqs = W.load_sequences(text)
    cat, offs = W.sample_reads(seqs, 8000, 150, seed=3, rc_frac=0.0)
    monkeypatch.setenv("PGX_SEED_K", "7")
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
    for min_len, min_occ in ((20, 1), (10, 1)):
        ref = O.find_mems_batch(ri, tags, cat, offs, min_len, min_occ, threads=O.lib().orc_max_threads())
        b = idx.batch(cat, offs)
        b.run(min_len, min_occ, P.RUN_TAGS | P.RUN_TIMING)
        assert b.timing().pairs_reads == 2  # packed reads, no text comparison
        _same(b.result(), ref)
        b.free()
    idx.close()

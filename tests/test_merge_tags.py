"""merge_tags: per-chromosome tag streams -> whole-genome tag array.

The reference holds no fixture for this path (it needs a GBZ), so the check is from first principles: every suffix
(sequence s, offset o) gets a tag g(s, o); the per-chromosome streams are g along each chromosome's own suffix array,
the expected result is g along the whole-genome suffix array, and the reference's sequential procedure
(merge_tags.cpp:289-405: walk the SA, take "the next tag" of the owning file) is restated in Python as the oracle."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W


def _g(seq, off):
    """tag of suffix (seq, off): node from a coarse offset bucket so that haplotype copies share tags and runs form"""
    node = 1 + (off // 5) + 1000 * (seq // 4)
    return (node << 11) | ((seq & 1) << 10) | (off % 5)


def _bytecode(v):
    out = bytearray()
    while v > 0x7F:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _write_algorithm_tags(path, tags, header):
    """build_tags' output: ByteCode runs offset:10 | rev:1 | len:9 | node << 20, runs of at most 511"""
    body = bytearray()
    i = 0
    while i < len(tags):
        j = i
        while j < len(tags) and tags[j] == tags[i] and j - i < 511:
            j += 1
        v = int(tags[i])
        body += _bytecode((v & 0x7FF) | ((j - i) << 11) | ((v >> 11) << 20))
        i = j
    with open(path, "wb") as f:
        if header:
            f.write(np.uint64(len(body) * 8).tobytes())  # int_vector<8> header of sdsl::int_vector_buffer<8>
        f.write(bytes(body))


def _setup(workdir, n_chrom=3, base_len=3000, tagf=None, pre="mt"):
    texts, seqs_all, chrom_of_seq, ri_c = [], [], [], []
    for c in range(n_chrom):
        t = os.path.join(workdir, "%s_chrom_%d.txt" % (pre, c))
        W.synth_pangenome_text(t, base_len=base_len + 700 * c, n_hap=2, seed=200 + c, n_runs=1, n_run_len=(20, 60))
        seqs = W.load_sequences(t)
        texts.append(t)
        ri_c.append(W.build_index_from_text(t, workdir, "%s_chrom_%d" % (pre, c), with_tags=False)[0])
        chrom_of_seq += [c] * len(seqs)
        seqs_all += seqs
    whole = os.path.join(workdir, "%s_whole.txt" % pre)
    with open(whole, "wb") as f:
        for s in seqs_all:
            f.write(bytes(s) + b"\n")
    ri_w = W.build_index_from_text(whole, workdir, "%s_whole" % pre, with_tags=False)[0]
    g = (lambda sq, off: _g(sq, off)) if tagf is None else (lambda sq, off: tagf(seqs_all, sq, off))
    # per-chromosome streams: g along the chromosome's own SA (non-endmarker positions), with GLOBAL sequence ids
    tag_paths, base = [], 0
    for c in range(n_chrom):
        r = O.RIndex(ri_c[c])
        sa, ml = r.decompress_sa(), r.max_length
        n_seq_c = sum(1 for x in chrom_of_seq if x == c)
        tags = [g(base + int(v) // ml, int(v) % ml) for v in sa[n_seq_c:]]
        p = os.path.join(workdir, "%s_chrom_%d.algo.tags" % (pre, c))
        _write_algorithm_tags(p, tags, header=(c % 2 == 0))  # both container flavours
        tag_paths.append(p)
        base += n_seq_c
    rw = O.RIndex(ri_w)
    sa, ml = rw.decompress_sa(), rw.max_length
    n_seq = len(seqs_all)
    expected = [0] * n_seq + [g(int(v) // ml, int(v) % ml) for v in sa[n_seq:]]
    return ri_w, tag_paths, np.array(chrom_of_seq, dtype=np.uint32), expected, rw


def _sequential_merge(rw, tag_paths, seq_to_file, n_seq):
    """the reference's procedure: BWT order, 'next tag' of the owning file's stream (merge_tags.cpp:322-333, 381-397)"""
    streams = []
    for p in tag_paths:
        raw = open(p, "rb").read()
        if len(raw) >= 8 and int(np.frombuffer(raw[:8], dtype=np.uint64)[0]) == (len(raw) - 8) * 8:
            raw = raw[8:]
        vals, i = [], 0
        while i < len(raw):
            v, sh = 0, 0
            while True:
                b = raw[i]; i += 1
                v |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            vals += [(v & 0x7FF) | ((v >> 20) << 11)] * ((v >> 11) & 0x1FF)
        streams.append(vals)
    cur = [0] * len(streams)
    out = [0] * n_seq
    sa, ml = rw.decompress_sa(), rw.max_length
    for v in sa[n_seq:]:
        f = int(seq_to_file[int(v) // ml])
        out.append(streams[f][cur[f]])
        cur[f] += 1
    assert cur == [len(s) for s in streams]
    return out


def _split511(runs):
    """append_compact_run_streamed (tag_arrays.cpp:940-974): uint16_t length, pieces of 511 while >= 512, nothing for 0"""
    out = []
    for v, ln in runs:
        assert 0 <= ln <= 0xFFFF
        while ln >= 512:
            out.append((v, 511)); ln -= 511
        if ln:
            out.append((v, ln))
    return out


def _maximal_runs(tags):
    runs, i = [], 0
    while i < len(tags):
        j = i
        while j < len(tags) and tags[j] == tags[i]:
            j += 1
        runs.append((tags[i], j - i))
        i = j
    return runs


def _reference_job_loop(tags, run_starts, n_seq, runs_per_job=500):
    """What the reference hands to append_compact_run_streamed, in order (merge_tags.cpp:597-823): the endmarker run and the rest of
    the BWT run that holds position n_seq first (:625,653-697), then jobs of 500 BWT runs (:600,733-823, extract_tags_batch :289-405),
    every count a uint16_t, the last run of a segment held back and added to the first run of the next when the tags are equal."""
    import bisect
    M, n = 0xFFFF, len(tags)
    tot = len(run_starts)
    out = []
    temp = [[0, n_seq & M]]                                   # :625 (size_t -> uint16_t)
    run_id = bisect.bisect_right(run_starts, n_seq) - 1       # run_id_and_offset_at(num_endmarkers, ..)
    end = run_starts[run_id + 1] if run_id + 1 < tot else n
    for p in range(n_seq, end):                               # :670-688
        if temp[-1][0] == tags[p]:
            temp[-1][1] = (temp[-1][1] + 1) & M
        else:
            temp.append([tags[p], 1])
    prev = temp.pop()                                         # :694
    out += [tuple(t) for t in temp]
    start = run_id + 1
    n_jobs = (tot - start + runs_per_job - 1) // runs_per_job  # :741
    for j in range(n_jobs):
        s0 = start + j * runs_per_job
        lo = run_starts[s0]
        hi = n if s0 + runs_per_job >= tot else run_starts[s0 + runs_per_job]  # :311-319 (the last job walks to the end of the BWT)
        cur = []
        for p in range(lo, hi):                               # :381-400
            if cur and cur[-1][0] == tags[p]:
                cur[-1][1] = (cur[-1][1] + 1) & M
            else:
                cur.append([tags[p], 1])
        if cur[0][0] == prev[0]:                              # :776-777
            cur[0][1] = (cur[0][1] + prev[1]) & M
        else:
            out.append(tuple(prev))                           # :779-782
        if j < n_jobs - 1:                                    # :812-815
            prev = cur.pop()
        out += [tuple(t) for t in cur]
    # (with no job at all the held-back run is never written: the reference loses it; not reached when position n_seq is not in the last run)
    return out


def _bwt_run_starts(rw, seqs):
    """run heads of the whole-genome BWT (one SA sample each, FastLocate::tot_runs): the symbol before each suffix; endmarkers are
    distinct symbols, so each is its own run"""
    sa, ml = rw.decompress_sa(), rw.max_length
    b = [seqs[int(v) // ml][int(v) % ml - 1] if int(v) % ml else 0 for v in sa]
    starts = [i for i in range(len(b)) if i == 0 or b[i] != b[i - 1] or b[i] == 0]
    assert len(starts) == rw.L.orc_ri_samples_size(rw.h)
    return starts


def test_reference_job_loop_gives_maximal_runs_mod_65536():
    """the reference's 500-run jobs and previous_last_run stitching leave no trace in the output -- its runs are
    the maximal runs, lengths mod 65 536 (what PGX_MERGE_REFERENCE_RUNS writes)"""
    rng = np.random.default_rng(77)
    for case in range(12):
        n_seq = int(rng.integers(1, 40)) if case % 3 else 65536 + 3
        lens = rng.geometric(0.3, size=int(rng.integers(800, 4000))).tolist()
        for L in ([65536, 70000, 131072 + 5, 65535, 512, 511, 1022, 1023][case % 8], 65536 * (case % 2) + 600):
            lens[int(rng.integers(0, len(lens)))] = L          # long merged runs that cross many jobs, and ones that wrap to 0
        vals = rng.integers(1, 50, size=len(lens))
        tags = [0] * n_seq
        for v, l in zip(vals, lens):
            tags += [int(v) << 11] * int(l)
        n = len(tags)
        # BWT run heads: random, dense enough that 500 of them are far shorter than the long tag runs; n_seq inside a run or at a head
        heads = sorted(set([0] + rng.integers(1, n, size=n // int(rng.integers(2, 40))).tolist() + ([n_seq] if case % 2 else [])))
        if heads[-1] <= n_seq:
            heads.append(n_seq + 1)                            # position n_seq must not lie in the last run (see _reference_job_loop)
        got = _reference_job_loop(tags, heads, n_seq)
        want = [(v, l & 0xFFFF) for v, l in _maximal_runs(tags)]
        assert _split511(got) == _split511(want), case
        assert [g for g in got if g[1]] == [w for w in want if w[1]], case


def test_merge_procedure_restated(workdir, built):
    ri_w, tag_paths, s2f, expected, rw = _setup(workdir)
    assert _sequential_merge(rw, tag_paths, s2f, len(s2f)) == expected


@pytest.mark.gpu
def test_merge_tags_gpu(workdir):
    ri_w, tag_paths, s2f, expected, rw = _setup(workdir)
    out = os.path.join(workdir, "mt_whole.tags")
    P.merge_tags(ri_w, tag_paths, s2f, out)
    t = O.Tags(out, O.TAGS_COMPACT)
    # expected runs: maximal, split at 511 (append_compact_run_streamed)
    runs, i = [], 0
    while i < len(expected):
        j = i
        while j < len(expected) and expected[j] == expected[i]:
            j += 1
        ln = j - i
        while ln >= 512:
            runs.append((expected[i], 511)); ln -= 511
        if ln:
            runs.append((expected[i], ln))
        i = j
    L = t.L
    assert L.orc_tags_n_runs(t.h) == len(runs) == L.orc_tags_n_items(t.h)
    pos = 0
    for k, (v, ln) in enumerate(runs):
        assert L.orc_tags_interval(t.h, k) == pos and L.orc_tags_item(t.h, k) == v, k
        pos += ln
    assert pos == rw.n
    # the merged file serves find_mems like any other tag array
    idx = P.Index(ri_w, out)
    seqs = W.load_sequences(os.path.join(workdir, "mt_whole.txt"))
    cat, offs = W.sample_reads(seqs, 2000, 100, seed=5)
    res = idx.find_mems(cat, offs, 15, 1, tags=True)
    ref = O.find_mems_batch(rw, t, cat, offs, 15, 1, threads=O.lib().orc_max_threads())
    assert res["mems"].tobytes() == ref["mems"].tobytes() and np.array_equal(res["positions"], ref["positions"])
    # the CLI (grouped sequences: --counts file:n,...) writes the same bytes
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(P.__file__)), "merge_tags")
    counts = ",".join("%s:%d" % (os.path.basename(tag_paths[c]), int((s2f == c).sum())) for c in range(len(tag_paths)))
    out2 = os.path.join(workdir, "mt_cli.tags")
    r = subprocess.run([exe, "--counts", counts, ri_w, workdir, "--out", out2], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Index files merged and ready to use!" in r.stderr, r.stderr
    assert open(out2, "rb").read() == open(out, "rb").read()
    mp = os.path.join(workdir, "mt_map.txt")
    open(mp, "w").write("".join(os.path.basename(tag_paths[int(c)]) + "\n" for c in s2f))
    r = subprocess.run([exe, mp, ri_w, workdir, "--out", out2 + "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and open(out2 + "2", "rb").read() == open(out, "rb").read(), r.stderr
    # a stream that does not match the index is rejected
    bad = tag_paths[:1] + tag_paths[:1] + tag_paths[2:]
    with pytest.raises(P.PgxError) as e:
        P.merge_tags(ri_w, bad, s2f, out + ".bad")
    assert e.value.code == P.ERR_FORMAT
    with pytest.raises(P.PgxError) as e:
        P.merge_tags(ri_w, tag_paths, s2f[:-1], out + ".bad")
    assert e.value.code == P.ERR_ARG


def _by_first_symbol(seqs, sq, off):
    """a tag that only depends on the suffix's first symbol: the whole A, C and G ranges of the BWT become one run"""
    c = seqs[sq][off] if off < len(seqs[sq]) else 0
    return (9 << 11) | 3 if c in b"ACG" else (17 << 11) | (1 << 10) | 5


def test_reference_job_loop_on_built_indexes(workdir, built):
    """the same rule with the run heads of real whole-genome BWTs (jobs of 500 of its runs)"""
    for kw in (dict(), dict(n_chrom=2, base_len=24000, tagf=_by_first_symbol, pre="mtl")):
        ri_w, tag_paths, s2f, expected, rw = _setup(workdir, **kw)
        seqs = W.load_sequences(os.path.join(workdir, "%s_whole.txt" % kw.get("pre", "mt")))
        heads = _bwt_run_starts(rw, seqs)
        assert len(heads) > 1500                              # several jobs
        loop = [r for r in _reference_job_loop(expected, heads, len(s2f)) if r[1]]
        mx = _maximal_runs(expected)
        assert loop == [(v, l & 0xFFFF) for v, l in mx if l & 0xFFFF]
        assert (max(l for _, l in mx) > 65536) == bool(kw)


@pytest.mark.gpu
def test_merge_tags_reference_runs(workdir):
    """--reference-runs / PGX_MERGE_REFERENCE_RUNS writes the file the reference's job loop leads to.  With a
    merged run of more than 65 535 positions the reference's uint16_t count wraps; without one both modes write the same bytes."""
    import subprocess
    # (a) ordinary tags: no long run, one file either way
    ri_w, tag_paths, s2f, expected, rw = _setup(workdir)
    a, b = os.path.join(workdir, "mt_exact.tags"), os.path.join(workdir, "mt_refruns.tags")
    P.merge_tags(ri_w, tag_paths, s2f, a)
    P.merge_tags(ri_w, tag_paths, s2f, b, flags=P.MERGE_REFERENCE_RUNS)
    assert open(a, "rb").read() == open(b, "rb").read()
    seqs = W.load_sequences(os.path.join(workdir, "mt_whole.txt"))
    loop = _reference_job_loop(expected, _bwt_run_starts(rw, seqs), len(s2f))
    assert _split511(loop) == _split511(_maximal_runs(expected))
    # (b) three quarters of the BWT under one tag
    ri_w, tag_paths, s2f, expected, rw = _setup(workdir, n_chrom=2, base_len=24000, tagf=_by_first_symbol, pre="mtl")
    seqs = W.load_sequences(os.path.join(workdir, "mtl_whole.txt"))
    assert max(l for _, l in _maximal_runs(expected)) > 65536
    loop = [r for r in _reference_job_loop(expected, _bwt_run_starts(rw, seqs), len(s2f)) if r[1]]
    assert loop == [(v, l & 0xFFFF) for v, l in _maximal_runs(expected) if l & 0xFFFF]
    a, b, c = (os.path.join(workdir, "mtl_%s.tags" % k) for k in ("exact", "refruns", "restated"))
    P.merge_tags(ri_w, tag_paths, s2f, a)
    P.merge_tags(ri_w, tag_paths, s2f, b, flags=P.MERGE_REFERENCE_RUNS)
    vals = np.array([v for v, _ in loop], dtype=np.uint64)
    lens = np.array([l for _, l in loop], dtype=np.uint64)
    P.write_compact_tags(c, vals, lens)
    assert open(b, "rb").read() == open(c, "rb").read() and open(a, "rb").read() != open(b, "rb").read()
    t = O.Tags(b, O.TAGS_COMPACT)
    pieces = _split511(loop)
    assert t.L.orc_tags_n_runs(t.h) == len(pieces)
    pos = 0
    for k, (v, ln) in enumerate(pieces):
        assert t.L.orc_tags_interval(t.h, k) == pos and t.L.orc_tags_item(t.h, k) == v, k
        pos += ln
    assert pos < rw.n                                       # the wrapped run lost positions, as in the reference
    # the exact file still covers the BWT
    t = O.Tags(a, O.TAGS_COMPACT)
    exact = sum((l + 510) // 511 if l >= 512 else 1 for _, l in _maximal_runs(expected))
    assert t.L.orc_tags_n_runs(t.h) == exact and sum(l for _, l in _maximal_runs(expected)) == rw.n
    # CLI switch
    exe = os.path.join(os.path.dirname(os.path.abspath(P.__file__)), "merge_tags")
    counts = ",".join("%s:%d" % (os.path.basename(tag_paths[f]), int((s2f == f).sum())) for f in range(len(tag_paths)))
    o2 = os.path.join(workdir, "mtl_cli.tags")
    r = subprocess.run([exe, "--counts", counts, ri_w, workdir, "--out", o2, "--reference-runs"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and open(o2, "rb").read() == open(b, "rb").read(), r.stderr
    with pytest.raises(P.PgxError) as e:
        P.merge_tags(ri_w, tag_paths, s2f, o2, flags=8)
    assert e.value.code == P.ERR_ARG

"""Synthetic workloads for tests and bench.py (SURVEY section 8d "Read synthesis" / configs).

No algorithm of the hot path lives here: this module only fabricates inputs (texts, reads, tag
runs) and drives the build-side entry points of the C ABI to obtain index files.
"""
import os
import struct

import numpy as np

import pgx_ffi
from pgx_ffi import build_rindex, build_rlbwt, write_compact_tags

_COMP = np.arange(256, dtype=np.uint8)
for a, b in (("A", "T"), ("C", "G"), ("G", "C"), ("T", "A")):
    _COMP[ord(a)] = ord(b)


def read_rlbwt_runs(path):
    """grlBWT run file -> (symbols uint8[], lengths uint64[])."""
    d = open(path, "rb").read()
    bs, bl = struct.unpack_from("<QQ", d, 0)
    rec = np.frombuffer(d, dtype=np.uint8, offset=16).reshape(-1, bs + bl)
    sym = rec[:, 0].copy()
    ln = np.zeros(len(rec), dtype=np.uint64)
    for b in range(bl):
        ln |= rec[:, bs + b].astype(np.uint64) << np.uint64(8 * b)
    return sym, ln


def logical_runs(sym, ln):
    """every endmarker is its own run (src/r-index.cpp:840-848)"""
    rep = np.where(sym == 10, ln, 1).astype(np.int64)
    s2 = np.repeat(sym, rep)
    l2 = np.repeat(np.where(sym == 10, 1, ln).astype(np.uint64), rep)
    return s2, l2


def synthetic_tags_from_runs(rlbwt_path, out_tags_path):
    """one tag run per logical BWT run, value = (run_id + 1) << 11."""
    sym, ln = read_rlbwt_runs(rlbwt_path)
    _, l2 = logical_runs(sym, ln)
    vals = (np.arange(len(l2), dtype=np.uint64) + np.uint64(1)) << np.uint64(11)
    write_compact_tags(out_tags_path, vals, l2)
    return len(l2)


def build_index_from_rlbwt(rlbwt_path, workdir, name, encoded=True, with_tags=True):
    os.makedirs(workdir, exist_ok=True)
    ri = os.path.join(workdir, name + (".ri" if encoded else ".legacy.ri"))
    build_rindex(rlbwt_path, ri, encoded)
    tags = None
    if with_tags:
        tags = os.path.join(workdir, name + ".compact.tags")
        synthetic_tags_from_runs(rlbwt_path, tags)
    return ri, tags


def build_index_from_text(text_path, workdir, name, encoded=True, with_tags=True, parts=None):
    """parts (None: by size): build through pgx_build_index_from_texts with the sequences of the text dealt into that many texts of consecutive
    sequences -- the same bytes as the single suffix array gives (tests/test_formats.py), in a fraction of the time on a many-core host"""
    os.makedirs(workdir, exist_ok=True)
    rl = os.path.join(workdir, name + ".rl_bwt")
    ri = os.path.join(workdir, name + (".ri" if encoded else ".legacy.ri"))
    if parts is None:
        parts = 8 if os.path.getsize(text_path) >= (64 << 20) else 1  # (chr22 workload on 8 cores: 166 s with four texts, 144 s with eight or sixteen)
    pieces = []
    if parts > 1:
        raw = open(text_path, "rb").read()
        ends = np.flatnonzero(np.frombuffer(raw, dtype=np.uint8) == 10) + 1  # one past every newline
        if len(raw) and raw[-1] != 10:
            ends = np.append(ends, len(raw))
        if len(ends) >= parts:
            cuts = [0] + [int(ends[len(ends) * (k + 1) // parts - 1]) for k in range(parts)]
            for k in range(parts):
                pth = os.path.join(workdir, "%s.part%d.txt" % (name, k))
                with open(pth, "wb") as f:
                    f.write(raw[cuts[k]:cuts[k + 1]])
                pieces.append(pth)
        del raw
    if pieces:
        pgx_ffi.build_index_from_texts(pieces, rl, ri, encoded)
        for pth in pieces:
            os.remove(pth)
    else:
        pgx_ffi.build_index_from_text(text_path, rl, ri, encoded)  # one suffix array for the BWT and the SA samples
    tags = None
    if with_tags:
        tags = os.path.join(workdir, name + ".compact.tags")
        synthetic_tags_from_runs(rl, tags)
    return ri, tags, rl


def build_index_from_texts(text_paths, workdir, name, encoded=True, with_tags=True):
    """the collection as several texts ("chromosomes"): pgx_build_index_from_texts (per-text suffix arrays + k-way merge; any total size)"""
    os.makedirs(workdir, exist_ok=True)
    rl = os.path.join(workdir, name + ".rl_bwt")
    ri = os.path.join(workdir, name + (".ri" if encoded else ".legacy.ri"))
    pgx_ffi.build_index_from_texts(list(text_paths), rl, ri, encoded)
    tags = None
    if with_tags:
        tags = os.path.join(workdir, name + ".compact.tags")
        synthetic_tags_from_runs(rl, tags)
    return ri, tags, rl


def _synth_one(args):
    path, kw = args
    return synth_pangenome_text(path, **kw)


def synth_chromosome_texts(workdir, name, n_chrom, base_len, n_hap, seed=45, processes=8, **kw):
    """n_chrom independent synthetic chromosomes (synth_pangenome_text each, its own seed), written side by side by worker processes"""
    import multiprocessing as mp

    paths = [os.path.join(workdir, "%s_chr%d.txt" % (name, c)) for c in range(n_chrom)]
    jobs = [(paths[c], dict(base_len=base_len, n_hap=n_hap, seed=seed + 1000 * c, **kw)) for c in range(n_chrom)]
    if processes > 1 and n_chrom > 1:
        with mp.get_context("fork").Pool(min(processes, n_chrom)) as pool:
            pool.map(_synth_one, jobs)
    else:
        for j in jobs:
            _synth_one(j)
    return paths


def load_sequences(text_path):
    raw = open(text_path, "rb").read()
    return [np.frombuffer(s, dtype=np.uint8) for s in raw.split(b"\n") if len(s)]


def sample_reads(seqs, n_reads, read_len=150, seed=42, sub_rate=0.01, rc_frac=0.5, n_frac=None):
    """150-bp reads sampled from the indexed text, 1 % substitutions, 50 % reverse-complemented.
    n_frac (None = wherever the uniform positions fall): the share of reads that overlap an N run of the text (their start is drawn
    around a uniformly chosen N position); the others are drawn again until they hold no N.

    Returns (reads uint8[n_reads * read_len], offsets uint64[n_reads + 1])."""
    rng = np.random.default_rng(seed)
    ok = [s for s in seqs if len(s) >= read_len]
    if not ok:
        raise ValueError("no sequence is long enough for the requested read length")
    lens = np.array([len(s) for s in ok], dtype=np.int64)
    cat = np.concatenate(ok)
    base = np.concatenate(([0], np.cumsum(lens)[:-1]))
    # positions with >= read_len symbols ahead, uniformly
    room = lens - read_len + 1
    which = rng.choice(len(ok), size=n_reads, p=room / room.sum())
    start = base[which] + (rng.random(n_reads) * room[which]).astype(np.int64)
    if n_frac is not None:
        isn = cat == ord("N")
        npos = np.flatnonzero(isn)
        ncum = np.concatenate(([0], np.cumsum(isn, dtype=np.int64)))
        ends = np.cumsum(lens)
        has_n = lambda st: (ncum[st + read_len] - ncum[st]) > 0  # noqa: E731
        want = rng.random(n_reads) < n_frac
        if want.any() and len(npos) == 0:
            raise ValueError("the text has no N to sample reads from")
        # reads over an N run: a start within read_len before a uniformly chosen N position, clipped to its sequence
        k = int(want.sum())
        if k:
            p = npos[rng.integers(0, len(npos), size=k)]
            st = p - rng.integers(0, read_len, size=k)
            sq = np.searchsorted(ends, p, side="right")
            start[want] = np.clip(st, base[sq], ends[sq] - read_len)
        # the others: drawn again while they hold an N (a few rounds)
        for _ in range(50):
            bad = ~want & has_n(start)
            if not bad.any():
                break
            w2 = rng.choice(len(ok), size=int(bad.sum()), p=room / room.sum())
            start[bad] = base[w2] + (rng.random(int(bad.sum())) * room[w2]).astype(np.int64)
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    step = max(1, (1 << 24) // read_len)
    ar = np.arange(read_len, dtype=np.int64)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for lo in range(0, n_reads, step):
        hi = min(n_reads, lo + step)
        r = cat[start[lo:hi, None] + ar[None, :]]
        sub = rng.random(r.shape) < sub_rate
        if sub.any():
            # uniform over the other three bases (bytes outside ACGT get any base)
            cur = r[sub]
            idx = np.searchsorted(acgt, cur)
            idx = np.where((idx < 4) & (acgt[np.minimum(idx, 3)] == cur), idx, rng.integers(0, 4, size=cur.shape))
            r = r.copy()
            r[sub] = acgt[(idx + rng.integers(1, 4, size=cur.shape)) % 4]
        rc = rng.random(hi - lo) < rc_frac
        if rc.any():
            r = r.copy() if not sub.any() else r
            r[rc] = _COMP[r[rc][:, ::-1]]
        out[lo:hi] = r
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return out.reshape(-1), offsets


def synth_pangenome_text(path, base_len=4_000_000, n_hap=8, seed=45, gc=0.41, snp=1e-3, indel=1e-4, n_runs=4,
                         n_run_len=(1000, 50000)):
    """SURVEY 8d config 3 recipe, scaled by base_len: i.i.d. base sequence with N runs, n_hap
    haplotypes with SNPs and indels, every haplotype in both orientations (so sigma = 6 and the
    index is a proper FMD index).  Writes newline-terminated sequences; returns their count."""
    rng = np.random.default_rng(seed)
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    basev = acgt[rng.choice(4, size=base_len, p=p)].copy()
    for _ in range(n_runs):
        ln = int(rng.integers(n_run_len[0], n_run_len[1]))
        ln = min(ln, base_len // (4 * n_runs) + 1)
        st = int(rng.integers(0, max(1, base_len - ln)))
        basev[st:st + ln] = ord("N")
    with open(path, "wb") as f:
        for _h in range(n_hap):
            hap = basev.copy()
            m = rng.random(base_len) < snp
            hap[m] = acgt[rng.integers(0, 4, size=int(m.sum()))]
            # indels: deletions drop 1..k bases, insertions add random bases
            nind = rng.binomial(base_len, indel)
            if nind:
                pos = np.sort(rng.choice(base_len, size=nind, replace=False))
                lens = rng.geometric(1 / 3.0, size=nind)
                isdel = rng.random(nind) < 0.5
                pieces, prev = [], 0
                for q, ln, dl in zip(pos, lens, isdel):
                    if q < prev:
                        continue
                    pieces.append(hap[prev:q])
                    if dl:
                        prev = min(base_len, q + int(ln))
                    else:
                        pieces.append(acgt[rng.integers(0, 4, size=int(ln))])
                        prev = q
                pieces.append(hap[prev:])
                hap = np.concatenate(pieces)
            f.write(hap.tobytes()); f.write(b"\n")
            f.write(_COMP[hap[::-1]].tobytes()); f.write(b"\n")
    return 2 * n_hap

"""The formulation behind pgx_find_mems_pairs_kernel<.., LCE> (pgx_image.h "LCE image"), CPU tier, oracle only: on a collection that holds every sequence
next to its reverse complement, the FORWARD stage of find_mems_function (algorithm.hpp:676-700: forward_extend until the interval is "small") ends where the
longest match of the read with the text behind any occurrence of the stage's first interval ends, and the interval it ends with is the (consecutive) run of
occurrences that reach that far: MEM end = j + max LCE, bwt_start = k + first such occurrence, size = their number.  Checked against the oracle's
find_mems_function on reads with substitutions, short reads, reads over N runs and sequence ends, for min_occ 1 (the only case the kernel takes this way).
Also: the text can be recovered from the index alone -- the first symbol of suffix i is the symbol whose C-bucket holds i (what pgx_lce_scatter_kernel does)."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_workload as W


def test_forward_stage_equals_longest_match_over_the_occurrences(workdir):
    text = os.path.join(workdir, "lce_math.txt")
    W.synth_pangenome_text(text, base_len=30_000, n_hap=5, seed=3, n_runs=3, n_run_len=(30, 300))
    ri_path = W.build_index_from_text(text, workdir, "lce_math", with_tags=False)[0]
    ri = O.RIndex(ri_path)
    seqs = W.load_sequences(text)
    ml = ri.max_length
    sa = ri.decompress_sa()
    seq_start = np.concatenate([[0], np.cumsum([len(s) + 1 for s in seqs])])
    T = np.concatenate([np.concatenate([s, [10]]) for s in seqs]).astype(np.uint8)
    assert len(T) == ri.n
    gpos = seq_start[(sa // ml).astype(np.int64)] + (sa % ml).astype(np.int64)  # the suffix array in text coordinates
    # first column: suffix i starts with the symbol whose bucket holds i (nuc order \n A C G N T)
    nuc = [10, 65, 67, 71, 78, 84]
    cum = np.concatenate([[0], np.cumsum([int((T == c).sum()) for c in nuc])])
    for b, c in enumerate(nuc):
        assert (T[gpos[cum[b]:cum[b + 1]]] == c).all()
    cat, offs = W.sample_reads(seqs, 1500, 150, seed=8, n_frac=0.05)
    rng = np.random.default_rng(1)
    checked = 0
    for r in range(1500):
        rd = bytes(cat[offs[r]:offs[r + 1]])
        if r % 5 == 0:
            rd = rd[: int(rng.integers(25, 150))]
        if r % 11 == 0:
            rd = bytes(seqs[r % len(seqs)][-len(rd):])  # a read that ends its sequence
        for min_len in (20, 12):
            for x in (0, int(rng.integers(0, max(1, len(rd) - min_len)))):
                nx, mem, _ = ri.find_mems_function(rd, min_len, 1, x)
                if mem is None or any(ch not in b"ACGT" for ch in rd[x:]):  # (reads with other bytes never reach the two-step kernel)
                    continue
                k, _, s = ri.bwd_pattern(rd[x:x + min_len])
                assert s > 0
                j = x + min_len
                rem = len(rd) - j
                best, first, cnt = -1, 0, 0
                for i in range(s):
                    p = int(gpos[k + i]) + min_len
                    l = 0
                    while l < rem and T[p + l] == rd[j + l]:
                        l += 1
                    if l > best:
                        best, first, cnt = l, i, 1
                    elif l == best:
                        cnt += 1
                assert (x, j + best, k + first, cnt) == tuple(mem), (rd, x, min_len)
                checked += 1
    assert checked > 3000


def _lcp_table(T, gpos, cap=254, unknown=255):
    """what pgx_lce_lcp_kernel writes: entry i = symbols suffix i shares with suffix i - 1 (capped), `unknown` where the words it looked at -- 64 symbols per round,
    five words of either suffix -- touch a 512-symbol line of the text that holds anything but A C G T (or lies behind the text)"""
    n = len(T)
    Tp = np.concatenate([T, np.full(2048, 1, dtype=np.uint8)])
    p, q = gpos[:-1].astype(np.int64), gpos[1:].astype(np.int64)
    l = np.zeros(n - 1, dtype=np.int64)
    alive = np.ones(n - 1, dtype=bool)
    for d in range(256):
        alive &= Tp[p + d] == Tp[q + d]
        l += alive
    rounds = np.minimum(4, l // 64 + 1)
    special = ~np.isin(Tp, np.frombuffer(b"ACGT", dtype=np.uint8))
    special[n:] = True
    line_bad = np.add.reduceat(special[: (len(Tp) // 512) * 512].astype(np.int64), np.arange(0, (len(Tp) // 512) * 512, 512)) > 0
    bad = np.zeros(n - 1, dtype=bool)
    for s0 in (p, q):
        w = s0 >> 4
        bad |= line_bad[w >> 5] | line_bad[(w + 4 * rounds) >> 5]
    out = np.where(bad, unknown, np.minimum(l, cap)).astype(np.int64)
    return np.concatenate([[0], out])


@pytest.mark.parametrize("haps", [5, 40])
def test_occurrences_after_the_first_follow_from_the_common_prefixes(workdir, haps):
    """the text path of pgx_find_mems_pairs_kernel<.., LCE> with img.lce_lcp, trip by trip: a COMPARE trip matches occurrence i with the text (l); a shorter match
    than the best so far ends the stage (the matches of sorted suffixes with one pattern rise, stay, fall -- never rise again); then, and in CHAIN trips, up to 16
    entries of the table: occurrence t matches min(previous, lcp[k + t] - m) -- m the symbols matched before the stage --, so an entry above the best adds one to the
    count, one below it ends the stage, one EQUAL to it (while the read goes on) or unknown asks for a COMPARE trip of occurrence t.  Same MEMs as the oracle,
    for intervals of up to 128 occurrences, in far fewer trips than occurrences."""
    text = os.path.join(workdir, "lce_math_%d.txt" % haps)
    W.synth_pangenome_text(text, base_len=800_000 // haps, n_hap=haps, seed=3 + haps, n_runs=3, n_run_len=(30, 300))
    ri_path = W.build_index_from_text(text, workdir, "lce_math_%d" % haps, with_tags=False)[0]
    ri = O.RIndex(ri_path)
    seqs = W.load_sequences(text)
    ml = ri.max_length
    sa = ri.decompress_sa()
    seq_start = np.concatenate([[0], np.cumsum([len(s) + 1 for s in seqs])])
    T = np.concatenate([np.concatenate([s, [10]]) for s in seqs]).astype(np.uint8)
    gpos = seq_start[(sa // ml).astype(np.int64)] + (sa % ml).astype(np.int64)
    lcp = _lcp_table(T, gpos)
    assert (lcp == 255).sum() < len(lcp) // 2  # (only near the N runs and the sequence ends)
    cat, offs = W.sample_reads(seqs, 1500, 150, seed=8, n_frac=0.05)
    rng = np.random.default_rng(1)
    checked = trips = occurrences = widest = 0
    for r in range(1500):
        rd = bytes(cat[offs[r]:offs[r + 1]])
        if r % 5 == 0:
            rd = rd[: int(rng.integers(25, 150))]
        if r % 11 == 0:
            rd = bytes(seqs[r % len(seqs)][-len(rd):])
        for min_len in (20, 12):
            for x in (0, int(rng.integers(0, max(1, len(rd) - min_len)))):
                nx, mem, _ = ri.find_mems_function(rd, min_len, 1, x)
                if mem is None or any(ch not in b"ACGT" for ch in rd[x:]):
                    continue
                k, _, s = ri.bwd_pattern(rd[x:x + min_len])
                if s > 128:
                    continue
                j, m = x + min_len, min_len
                rem = len(rd) - j
                best, first, cnt, i, compare = -1, 0, 0, 0, True
                while True:  # one trip
                    trips += 1
                    done = False
                    if compare:
                        p = int(gpos[k + i]) + m
                        l = 0
                        while l < rem and T[p + l] == rd[j + l]:
                            l += 1
                        if i == 0 or l > best:
                            best, first, cnt = l, i, 1
                        elif l == best:
                            cnt += 1
                        else:
                            break  # shorter than the best: so is everything behind it
                        t = i + 1
                    else:
                        t = i
                    t0, compare = t, True
                    while t < s and t < t0 + 16:
                        c = int(lcp[k + t])
                        if c == 255 or c < m or (c - m == best and best < rem):
                            break  # occurrence t is compared itself
                        if c - m < best:
                            done = True
                            break
                        cnt += 1
                        t += 1
                    else:
                        if t >= s:
                            done = True
                        else:
                            compare = False  # the window is used up: on with the next sixteen entries
                    if done:
                        break
                    i = t
                assert (x, j + best, k + first, cnt) == tuple(mem), (rd, x, min_len)
                checked += 1
                occurrences += s
                widest = max(widest, s)
    assert checked > 3000 and widest > (16 if haps > 16 else 4)
    print("haplotypes %d: stages %d, occurrences %d (widest interval %d), trips %d" % (haps, checked, occurrences, widest, trips))
    assert trips < 0.6 * occurrences

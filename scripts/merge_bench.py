"""pgx_merge_tags at the size of the synthetic pangenome (GPU box): python3 -u scripts/merge_bench.py [groups]
The whole-genome index is the bench's synthetic pangenome (n = 64 M, 16 sequences); the sequences are split into `groups`
"chromosomes"; each group's tag stream is a position-derived tag function along that group's own BWT order."""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pgx_ffi as P, pgx_workload as W

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wd = "/tmp/pgx_merge_bench"; os.makedirs(wd, exist_ok=True)
text = os.path.join(wd, "t.txt")
W.synth_pangenome_text(text)
ri = W.build_index_from_text(text, wd, "t", with_tags=False)[0]
idx = P.Index(ri)
inf = idx.info()
n, n_seq, ml = int(inf.bwt_size), int(inf.n_sequences), int(inf.max_length)
sa = idx.decompress_sa()
seq, off = (sa // np.uint64(ml)).astype(np.int64), (sa % np.uint64(ml)).astype(np.int64)
tags = (((off // 32 + 1) << 11) | (off % 32)).astype(np.uint64)  # node = offset / 32 + 1: haplotype copies share tags
s2f = (np.arange(n_seq) * K // n_seq).astype(np.uint32)
file_of = s2f[seq]
paths = []
for f in range(K):
    t = tags[n_seq:][file_of[n_seq:] == f]
    head = np.flatnonzero(np.concatenate(([True], t[1:] != t[:-1])))
    ln = np.diff(np.concatenate((head, [len(t)])))
    vals, lens = [], []
    for v, l in zip(t[head].tolist(), ln.tolist()):  # split runs at 511
        while l >= 512:
            vals.append(v); lens.append(511); l -= 511
        if l:
            vals.append(v); lens.append(l)
    v = np.array(vals, dtype=np.uint64); l = np.array(lens, dtype=np.uint64)
    d = (v & np.uint64(0x7FF)) | (l << np.uint64(11)) | ((v >> np.uint64(11)) << np.uint64(20))
    # ByteCode: 7 bits per byte, low group first
    nb = np.maximum(1, (np.floor(np.log2(np.maximum(d, 1).astype(np.float64))).astype(np.int64) // 7) + 1)
    nb = np.where(d >> (np.uint64(7) * nb.astype(np.uint64)) > 0, nb + 1, nb)  # guard float rounding
    out = np.zeros(int(nb.sum()), dtype=np.uint8)
    pos = np.concatenate(([0], np.cumsum(nb)[:-1]))
    for k in range(int(nb.max())):
        m = nb > k
        b = ((d[m] >> np.uint64(7 * k)) & np.uint64(0x7F)).astype(np.uint8)
        b |= np.where(nb[m] > k + 1, 0x80, 0).astype(np.uint8)
        out[pos[m] + k] = b
    p = os.path.join(wd, "group_%d.tags" % f)
    with open(p, "wb") as fh:
        fh.write(np.uint64(len(out) * 8).tobytes()); fh.write(out.tobytes())
    paths.append(p)
    print("group %d: %d tags in %d runs, %.1f MB" % (f, len(t), len(v), len(out) / 1e6), flush=True)
outp = os.path.join(wd, "whole.tags")
for rep in range(2):
    t0 = time.time(); P.merge_tags(ri, paths, s2f, outp); dt = time.time() - t0
    print("pgx_merge_tags: n = %d, %d files -> %.2f s wall (index load + image + DA + merge + write), output %.1f MB" % (n, K, dt, os.path.getsize(outp) / 1e6), flush=True)
# check against the direct run-length encoding of the tag function
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O
tg = O.Tags(outp, O.TAGS_COMPACT)
exp = tags.copy(); exp[:n_seq] = 0
head = np.flatnonzero(np.concatenate(([True], exp[1:] != exp[:-1])))
print("runs in the merged file:", tg.L.orc_tags_n_runs(tg.h), " maximal runs expected:", len(head), "(the file splits runs at 511)", flush=True)
assert tg.L.orc_tags_n_runs(tg.h) == len(head)  # no run reaches 512 with this tag function
for k in np.random.default_rng(1).integers(0, len(head), 5000):
    k = int(k)
    assert tg.L.orc_tags_interval(tg.h, k) == int(head[k]) and tg.L.orc_tags_item(tg.h, k) == int(exp[head[k]]), k
print("5000 random runs agree (start and value)")

import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pgx_workload as W, pgx_ffi as P, oracle_ffi as O
wd = tempfile.mkdtemp()
text = os.path.join(wd, "w.txt")
W.synth_pangenome_text(text, base_len=30_000, n_hap=3, seed=77, snp=0.01, indel=0.001, n_runs=3, n_run_len=(20, 400))
ri, tags, _ = W.build_index_from_text(text, wd, "w")
o = O.RIndex(ri)
os.environ["PGX_SB_SHIFT"] = "22"
idx = P.Index(ri, None, mode=P.MODE_IMAGE_DENSE2 | P.MODE_IMAGE_WIDE)
n = o.n
exp = np.array([o.rank6_true(p) for p in range(n + 1)], dtype=np.uint64)
for trial in range(3):
    pos = np.arange(0, n + 1, dtype=np.uint64)
    got = idx.rank_batch(pos, true_codes=True)
    bad = np.flatnonzero((got != exp).any(axis=1))
    print("trial", trial, "all positions: bad", len(bad), "first", bad[:5], "last", bad[-5:] if len(bad) else "")
    if len(bad):
        blks = np.unique(bad // 384)
        print("   bad blocks", len(blks), blks[:20], "subs of bad", np.bincount((bad % 384) >> 7, minlength=3))
        b = int(blks[0])
        sel = bad[bad // 384 == b]
        print("   block", b, "bad rels", (sel % 384)[:10], "...", (sel % 384)[-5:], "delta sample", (got[sel[0]].astype(np.int64) - exp[sel[0]].astype(np.int64)))
# a small batch (one workgroup) of high positions, and single positions
for cnt in (64, 256, 1024, 4096, 65536):
    pos = np.arange(n - cnt, n, dtype=np.uint64)
    got = idx.rank_batch(pos, true_codes=True)
    print("last", cnt, "positions: bad", int((got != exp[n - cnt:n]).any(axis=1).sum()))
pos = np.arange(0, n + 1, dtype=np.uint64)[::-1].copy()
got = idx.rank_batch(pos, true_codes=True)
bad = np.flatnonzero((got != exp[::-1]).any(axis=1))
print("reversed order: bad", len(bad), "positions", pos[bad][:5] if len(bad) else "")

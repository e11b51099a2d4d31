"""decompressSA / locate throughput on the synthetic pangenome (GPU box): python3 scripts/locate_bench.py [base_len]"""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pgx_ffi as P, pgx_workload as W, oracle_ffi as O

base_len = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
wd = "/tmp/pgx_locate_bench"; os.makedirs(wd, exist_ok=True)
text = os.path.join(wd, "t.txt")
W.synth_pangenome_text(text, base_len=base_len)
ri = W.build_index_from_text(text, wd, "t", with_tags=False)[0]
idx = P.Index(ri)
inf = idx.info()
print("n = %d, runs = %d" % (inf.bwt_size, inf.n_samples), flush=True)
idx.decompress_sa()  # upload of the locate image + warm-up
for rep in range(2):
    t0 = time.time(); sa = idx.decompress_sa(); dt = time.time() - t0
    print("GPU decompressSA: %.1f ms wall incl. %.0f MB device->host (%.2f G suffixes/s)" % (dt * 1e3, sa.nbytes / 1e6, len(sa) / dt / 1e9), flush=True)
rng = np.random.default_rng(1)
first = rng.integers(0, inf.bwt_size - 64, 2_000_000).astype(np.uint64)
last = first + rng.integers(0, 32, len(first)).astype(np.uint64)
for flags, name in ((0, "SA values"), (P.LOCATE_SEQ_IDS | P.LOCATE_UNIQUE, "sorted unique sequence ids")):
    idx.locate_batch(first[:1000], last[:1000], flags)
    t0 = time.time(); off, vals = idx.locate_batch(first, last, flags); dt = time.time() - t0
    print("GPU locate, 2 M ranges (mean 16.5 positions), %s: %.1f ms wall (%.1f M ranges/s, %d values)" % (name, dt * 1e3, len(first) / dt / 1e6, len(vals)), flush=True)
r = O.RIndex(ri)
m = 2_000_000
t0 = time.time()
L = r.L
v = L.orc_locate_first(r.h)
out = np.zeros(m, dtype=np.uint64)
L.orc_locate_sa(r.h, O.MODE_STRICT, 0, m - 1, out.ctypes.data)
dt = time.time() - t0
print("CPU oracle chain (1 thread): %d locateNext steps in %.2f s = %.2f M/s -> %.1f s for the whole SA" % (m, dt, m / dt / 1e6, inf.bwt_size / (m / dt)), flush=True)
assert np.array_equal(out, sa[:m])

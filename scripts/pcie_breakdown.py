"""PCIe-inclusive rate of the host-buffer API at the C level (what a C++ caller such as the find_mems CLI
pays): pgx_batch_upload (H2D of reads + offsets) + pgx_batch_run + pgx_batch_result (D2H into pinned buffers),
on a long-lived batch.  Python-side array copies are excluded."""
import sys, os, time
sys.path.insert(0, 'pangenome-index_amd'); sys.path.insert(0, 'tests')
import numpy as np, ctypes as C
import pgx_ffi as P, pgx_workload as W

wl = sys.argv[1] if len(sys.argv) > 1 else 'x'
golden = 'tests/golden'
if wl == 'x':
    ri, tags = W.build_index_from_rlbwt(golden + '/x.rl_bwt', '/tmp/pcie_wd', 'x')
    seqs = W.load_sequences(golden + '/x.newline_separated'); ml = 10
else:
    W.synth_pangenome_text('/tmp/pcie_wd_s.txt', base_len=4_000_000, n_hap=8, seed=45)
    ri, tags, _ = W.build_index_from_text('/tmp/pcie_wd_s.txt', '/tmp/pcie_wd', 's')
    seqs = W.load_sequences('/tmp/pcie_wd_s.txt'); ml = 20
cat, offs = W.sample_reads(seqs, 1_000_000, 150, seed=44)
idx = P.Index(ri, tags); idx.to_device(0)
L = idx.L
b = P.Batch(idx, cat, offs, 0)
r = P.Result()
for rep in range(4):
    t0 = time.perf_counter(); b.upload(cat, offs); t1 = time.perf_counter()
    b.run(ml, 1, P.RUN_TAGS); t2 = time.perf_counter()
    P._check(L.pgx_batch_result(b.b, C.byref(r))); t3 = time.perf_counter()
    print('%s: upload %.1f ms  run %.1f ms  result %.1f ms  -> %.1f M reads/s PCIe-inclusive (%d MEMs, %d positions)' % (
        wl, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1.0 / (t3 - t0), r.n_mems, r.n_positions))

#!/bin/bash
# round 4, experiment 1: (a) two-line probes by where the second line is, (b) fresh-batch rate against the number of workers, (c) does lower occupancy of
# the pairs kernel let a second batch's small kernels run under it
set -e
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather_pairs.hip -o /tmp/ubench_pairs
/tmp/ubench_pairs > gpurun_out/r4_ubench_pairs.txt
cat gpurun_out/r4_ubench_pairs.txt
W=/tmp/pgxwd; mkdir -p $W
for w in 3 5; do
  python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-overlap --no-parity --fresh-workers $w --steps 20 > gpurun_out/r4_fresh_w$w.json 2> gpurun_out/r4_fresh_w$w.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_fresh_w$w.json").read().strip().splitlines()[-1])
f=d["fresh_batch"]
print("workers $w: resident %.1f M/s (%.2f ms); fresh packed %.1f M/s (%.2f ms) %s; bytes %.1f M/s (%.2f ms) %s" % (d["value"]/1e6, d["ms_per_step"], f["packed"]["value"]/1e6, f["packed"]["ms_per_step"], f["packed"]["host_ms_per_step_inside"], f["bytes"]["value"]/1e6, f["bytes"]["ms_per_step"], f["bytes"]["host_ms_per_step_inside"]))
PY
done
for wg in 5 4 3; do
  PGX_FM_WG_PER_CU=$wg python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-parity --no-fresh --overlap --steps 20 > gpurun_out/r4_ov_wg$wg.json 2> gpurun_out/r4_ov_wg$wg.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_ov_wg$wg.json").read().strip().splitlines()[-1])
print("wg/cu $wg: one batch %.1f M/s (%.2f ms, main %.2f); two in flight %.1f M/s (%.2f ms)" % (d["value"]/1e6, d["ms_per_step"], d["kernel_ms_per_step"]["find_mems_main"], d["two_batches_in_flight"]["value"]/1e6, d["two_batches_in_flight"]["ms_per_step"]))
PY
done

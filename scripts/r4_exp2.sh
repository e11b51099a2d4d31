#!/bin/bash
# round 4, experiment 2: haplotype sweep (interval width), heavy threshold of the side launch, fresh-batch rate after the host-side trim
set -e
mkdir -p gpurun_out
bash scripts/haps_sweep.sh 32 64 96 2>&1 | grep -E "^haps|rror" || true
W=/tmp/pgxwd; mkdir -p $W
for hx in 2048 512 128; do
  PGX_FM_SIDE_HEAVY_EXT=$hx python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-parity --no-fresh --steps 20 > gpurun_out/r4_side_$hx.json 2> gpurun_out/r4_side_$hx.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_side_$hx.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("side heavy_ext $hx: %.1f M/s (%.2f ms): find_mems %.2f, main %.2f, behind it %.2f; compact %.2f tags %.2f" % (d["value"]/1e6, d["ms_per_step"], k["find_mems"], k["find_mems_main"], k["find_mems"]-k["find_mems_main"], k["compact"], k["tag_locate"]+k["tag_gather"]+k["tag_sort"]))
PY
done
for w in 3 4; do
  python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-parity --fresh-workers $w --steps 20 > gpurun_out/r4_fresh2_w$w.json 2> gpurun_out/r4_fresh2_w$w.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_fresh2_w$w.json").read().strip().splitlines()[-1])
f=d["fresh_batch"]
print("workers $w: resident %.1f M/s (%.2f ms); fresh packed %.1f M/s (%.2f ms) %s; bytes %.1f M/s (%.2f ms) %s" % (d["value"]/1e6, d["ms_per_step"], f["packed"]["value"]/1e6, f["packed"]["ms_per_step"], f["packed"]["host_ms_per_step_inside"], f["bytes"]["value"]/1e6, f["bytes"]["ms_per_step"], f["bytes"]["host_ms_per_step_inside"]))
PY
done

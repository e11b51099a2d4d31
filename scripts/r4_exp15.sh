#!/bin/bash
# round 4, experiment 15: how many workgroups the second-stream launch (reads with a byte outside ACGT, one-step kernel) takes next to the pairs kernel
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
B="python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-parity --no-fresh --steps 10"
run() {
  env $2 $B $3 > gpurun_out/r4_x15_$1.json 2> gpurun_out/r4_x15_$1.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_x15_$1.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("$1: step %.2f ms, find_mems %.2f (main %.2f), %.1f M reads/s" % (d["ms_per_step"], k["find_mems"], k["find_mems_main"], d["value"]/1e6))
PY
}
run base A=1
run min32_1 "PGX_FM_SIDE_WGS_MIN=32 PGX_FM_SIDE_PER_LANE=1"
run min32_2 "PGX_FM_SIDE_WGS_MIN=32 PGX_FM_SIDE_PER_LANE=2"
run min32_4 "PGX_FM_SIDE_WGS_MIN=32 PGX_FM_SIDE_PER_LANE=4"
run base2 A=1
run n0 A=1 "--n-read-frac 0"
run f2_base A=1 "--n-read-frac 0.02"
run f2_min32_4 "PGX_FM_SIDE_WGS_MIN=32 PGX_FM_SIDE_PER_LANE=4" "--n-read-frac 0.02"
run f2_min32_8 "PGX_FM_SIDE_WGS_MIN=32 PGX_FM_SIDE_PER_LANE=8" "--n-read-frac 0.02"

#!/bin/bash
# round 4, experiment 16: the second-stream launch in front of the pairs kernel (serial) instead of next to it; its grid
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
B="python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10"
run() {
  env $2 $B $3 > gpurun_out/r4_x16_$1.json 2> gpurun_out/r4_x16_$1.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_x16_$1.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("$1: step %.2f ms, find_mems %.2f (main %.2f), %.1f M reads/s, parity %s" % (d["ms_per_step"], k["find_mems"], k["find_mems_main"], d["value"]/1e6, d["parity_sample"]["identical"]))
PY
}
run base A=1
run serial PGX_FM_SIDE_SERIAL=1
run serial_4percu "PGX_FM_SIDE_SERIAL=1 PGX_FM_SIDE_WGS_MIN=1024"
run serial_2percu "PGX_FM_SIDE_SERIAL=1 PGX_FM_SIDE_WGS_MIN=512"
run base2 A=1

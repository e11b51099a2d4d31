#!/bin/bash
# round 4, experiment 5: the text-comparison path (LCE): widest interval that takes it, workgroups per CU, lane statistics
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-parity --no-fresh --steps 10 > gpurun_out/r4_lce_$name.json 2> gpurun_out/r4_lce_$name.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_lce_$name.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]
print("$name: %.1f M reads/s, step %.2f ms, main %.2f ms, lines %.1f M, seeds %.1f M, frac %.3f, lines/s %.1f G" % (d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], r["probes_issued"]/1e6, r["seed_loads"]/1e6, r["frac"], r["lines_per_s"]/1e9))
PY
}
run off PGX_FM_LCE=0
run max16 PGX_FM_LCE_MAX=16
run max8 PGX_FM_LCE_MAX=8
run max12 PGX_FM_LCE_MAX=12
run max24 PGX_FM_LCE_MAX=24
run wg3 PGX_FM_WG_PER_CU=3
run wg2 PGX_FM_WG_PER_CU=2
WLS=chr22 bash scripts/fm_stats.sh --workdir $W 2>&1 | tail -4

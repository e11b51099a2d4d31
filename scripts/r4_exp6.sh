#!/bin/bash
# round 4, experiment 6: how many idle lanes a wave of the LCE kernel gathers before it fetches new reads
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10 > gpurun_out/r4_lce_$name.json 2> gpurun_out/r4_lce_$name.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_lce_$name.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]
print("$name: %.1f M reads/s, step %.2f ms, main %.2f ms, lines %.1f M, frac %.3f, lines/s %.1f G, parity %s" % (d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], r["probes_issued"]/1e6, r["frac"], r["lines_per_s"]/1e9, d["parity_sample"]["identical"]))
PY
}
for m in 1 3 6 10 16 24; do run refill$m PGX_FM_REFILL_MIN=$m; done

#!/bin/bash
# round 4, experiment 9b: default workload with the common-prefix table (default) and without it (PGX_FM_LCP=0)
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
for t in lcp nolcp; do
[ $t = nolcp ] && export PGX_FM_LCP=0
python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10 > gpurun_out/r4_lcp_$t.json 2> gpurun_out/r4_lcp_$t.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r4_lcp_$t.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]
print("$t: %.1f M reads/s, step %.2f ms, main %.2f ms, lines %.1f M, seeds %.1f M, frac %.3f, parity %s" % (d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], r["probes_issued"]/1e6, r["seed_loads"]/1e6, r["frac"], d["parity_sample"]["identical"]))
PY
done

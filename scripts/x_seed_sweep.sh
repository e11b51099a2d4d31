#!/bin/bash
# usage (GPU box): bash scripts/x_seed_sweep.sh "<K> <K> ..."   -- bench.py --workload x once per seed depth (PGX_SEED_K)
export TMPDIR=/tmp
R=$PWD/gpurun_out/xseed; mkdir -p $R
for K in $1; do
  PGX_SEED_K=$K python3 bench.py --workload x --steps 20 --warmup 3 --no-secondary --no-cpu-baseline > $R/k$K.json 2> $R/k$K.err || echo "FAIL K=$K"
  python3 - $K $R/k$K.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernel_ms_per_step"]
print("K=%-3s fm %.3f ms  step %.3f ms  %.1f M reads/s" % (sys.argv[1], k["find_mems"], d["ms_per_step"], d["value"] / 1e6), flush=True)
PY
done

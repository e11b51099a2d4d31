#!/bin/bash
# usage (GPU box): bash scripts/synth_sweep.sh <tag> "<ENV=.. ENV=..>" ...   -- bench.py --workload synth once per environment setting
tag=$1; shift
export TMPDIR=/tmp
R=$PWD/gpurun_out/ssweep_$tag; mkdir -p $R
i=0
for E in "$@"; do
  i=$((i+1))
  env $E python3 bench.py --workload synth --workdir /tmp/wds --steps 10 --warmup 2 --no-secondary --no-cpu-baseline > $R/run$i.json 2> $R/run$i.err || echo "FAIL run $i: $E"
  python3 - "$E" $R/run$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernel_ms_per_step"]
print("%-40s fm %.3f ms  step %.3f ms  %.1f M reads/s  image %s" % (sys.argv[1], k["find_mems"], d["ms_per_step"], d["value"] / 1e6, d["config"].get("rank_image")), flush=True)
PY
done

#!/bin/bash
# usage (GPU box): bash scripts/chr22_sweep.sh <tag> "<ENV=.. ENV=..>" ["<ENV..>" ...]
# one chr22-scale index build (cached in /tmp/wd for the call), then bench.py once per environment setting; prints a table
tag=$1; shift
export TMPDIR=/tmp
R=$PWD/gpurun_out/sweep_$tag; mkdir -p $R
i=0
for E in "$@"; do
  i=$((i+1))
  env $E python3 bench.py --workdir /tmp/wd --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > $R/run$i.json 2> $R/run$i.err || echo "FAIL run $i: $E"
  python3 - "$E" $R/run$i.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["kernel_ms_per_step"]
    print("%-60s fm %.2f ms  step %.2f ms  %.1f M reads/s  frac %.3f  tags %.2f" % (sys.argv[1], k["find_mems"], d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"],
          k["tag_locate"] + k["tag_gather"] + k["tag_sort"]), flush=True)
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
done

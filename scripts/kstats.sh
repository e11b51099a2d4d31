#!/bin/bash
# usage (GPU box): bash scripts/kstats.sh <tag> [bench.py arguments...]   -- rocprofv3 kernel statistics of one bench.py run (environment as exported by the caller)
tag=$1; shift
export TMPDIR=/tmp
R=$PWD/gpurun_out/kstats_$tag; mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/out -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workdir /tmp/wd "$@" > $R/bench.json 2> $R/err.txt || echo FAIL
python3 - $R <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/out/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) >= 0.3:
            print("%-110s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY

#!/bin/bash
# usage (GPU box): bash scripts/pmc_pass.sh <tag> <workload> <counter> [<counter>...]
# One rocprofv3 --pmc pass per counter (kernel trace only, no other trace domain) over a short bench run; prints, per counter, the
# average value over the launches of the find_mems kernels.
tag=$1; wl=$2; shift 2
export TMPDIR=/tmp
R=$PWD/gpurun_out/pmc_$tag; mkdir -p $R
for C in "$@"; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$C -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-parity --no-fresh --workdir /tmp/wd > $R/$C.json 2> $R/$C.err || echo FAIL $C
done
python3 - $R "$@" <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
for C in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/%s/*/*_counter_collection.csv" % (R, C)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "find_mems" in k:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print("%-70s %-22s launches %3d  mean %.6g  max %.6g" % (k[:70], c, len(v), sum(v) / len(v), max(v)))
PY

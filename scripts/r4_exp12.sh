#!/bin/bash
# round 4, experiment 12: the text path for intervals of up to 128 occurrences (sixteen entries of the table of common prefixes per trip, suffix array entries
# fetched one by one): parity tests of the kernel, the default workload, 32 and 64 haplotypes at the same n
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
python -m pytest tests/test_gpu_pairs.py tests/test_gpu_parity.py tests/test_wide_image.py -m gpu -x -q > gpurun_out/r4_x12_tests.log 2>&1 || { tail -30 gpurun_out/r4_x12_tests.log; exit 1; }
tail -2 gpurun_out/r4_x12_tests.log
show() {
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_x12_$1.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]
print("$1: %.1f M reads/s, step %.2f ms, main %.2f ms, post %.2f ms, lines %.1f M, seeds %.1f M, parity %s" % (d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], k["total"]-k["find_mems"], r["probes_issued"]/1e6, r["seed_loads"]/1e6, d["parity_sample"]["identical"]))
PY
}
B="python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10"
$B > gpurun_out/r4_x12_base.json 2> gpurun_out/r4_x12_base.err; show base
for h in 32 64; do
  WH=/tmp/pgxwd_h$h; mkdir -p $WH
  BH="python bench.py --workdir $WH --haps $h --base-len $((320000000 / h)) --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10"
  $BH > gpurun_out/r4_x12_h$h.json 2> gpurun_out/r4_x12_h$h.err; show h$h
  PGX_FM_LCE_MAX=16 $BH > gpurun_out/r4_x12_h${h}_max16.json 2> gpurun_out/r4_x12_h${h}_max16.err; show h${h}_max16
done

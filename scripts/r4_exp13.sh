#!/bin/bash
# round 4, experiment 13: two first steps per trip (a stage that ends in its first step starts the next one at once; its seed entry arrives with the trip's
# lines): parity tests of the kernel, then the default workload against a build with one first step per trip
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
python -m pytest tests/test_gpu_pairs.py tests/test_gpu_parity.py tests/test_wide_image.py -m gpu -x -q > gpurun_out/r4_x13_tests.log 2>&1 || { tail -30 gpurun_out/r4_x13_tests.log; exit 1; }
tail -2 gpurun_out/r4_x13_tests.log
show() {
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_x13_$1.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]
print("$1: %.1f M reads/s, step %.2f ms, main %.2f ms, lines %.1f M, seeds %.1f M, parity %s" % (d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], r["probes_issued"]/1e6, r["seed_loads"]/1e6, d["parity_sample"]["identical"]))
PY
}
B="python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10"
$B > gpurun_out/r4_x13_two.json 2> gpurun_out/r4_x13_two.err; show two
D=/tmp/pgx_one; rm -rf $D; mkdir -p $D
cp -r pangenome-index_amd include oracle bench.py __graft_entry__.py tests $D/
(cd $D/pangenome-index_amd && rm -rf build libpgx.so && make -s -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -DPGX_FUSE2=0" libpgx.so)
(cd $D && $B) > gpurun_out/r4_x13_one.json 2> gpurun_out/r4_x13_one.err; show one
$B > gpurun_out/r4_x13_two2.json 2> gpurun_out/r4_x13_two2.err; show two2
WLS=chr22 bash scripts/fm_stats.sh --workdir /tmp/wd 2>&1 | tail -2

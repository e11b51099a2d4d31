#!/bin/bash
# round 4, experiment 7: the LCE kernel bounded to five waves per SIMD (96 VGPRs, spills) against four (110 VGPRs); seed table of depth 16 (64 GiB)
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
show() {
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_lce_$1.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]
print("$1: %.1f M reads/s, step %.2f ms, main %.2f ms, lines %.1f M, seeds %.1f M, frac %.3f, parity %s" % (d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], r["probes_issued"]/1e6, r["seed_loads"]/1e6, r["frac"], d["parity_sample"]["identical"]))
PY
}
python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10 > gpurun_out/r4_lce_base4.json 2> gpurun_out/r4_lce_base4.err; show base4
D=/tmp/pgx_w5; rm -rf $D; mkdir -p $D
cp -r pangenome-index_amd include oracle bench.py __graft_entry__.py tests $D/
(cd $D/pangenome-index_amd && rm -rf build libpgx.so && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -DPGX_LCE_WAVES=5" libpgx.so)
(cd $D && python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10) > gpurun_out/r4_lce_waves5.json 2> gpurun_out/r4_lce_waves5.err; show waves5
PGX_SEED_K=16 python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10 > gpurun_out/r4_lce_k16.json 2> gpurun_out/r4_lce_k16.err; show k16

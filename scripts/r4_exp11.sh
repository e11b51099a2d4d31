#!/bin/bash
# round 4, experiment 11: refill batching, widest interval and entry rule of the text path under the common-prefix table (default workload, same box)
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
show() {
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_x11_$1.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]
print("$1: %.1f M reads/s, step %.2f ms, main %.2f ms, lines %.1f M, parity %s" % (d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], r["probes_issued"]/1e6, d["parity_sample"]["identical"]))
PY
}
B="python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10"
$B > gpurun_out/r4_x11_base.json 2> gpurun_out/r4_x11_base.err; show base
for v in 4 8 20 32; do PGX_FM_REFILL_MIN=$v $B > gpurun_out/r4_x11_refill$v.json 2> gpurun_out/r4_x11_refill$v.err; show refill$v; done
PGX_FM_LCE_MAX=8 $B > gpurun_out/r4_x11_lcemax8.json 2> gpurun_out/r4_x11_lcemax8.err; show lcemax8
D=/tmp/pgx_cap16; rm -rf $D; mkdir -p $D
cp -r pangenome-index_amd include oracle bench.py __graft_entry__.py tests $D/
(cd $D/pangenome-index_amd && rm -rf build libpgx.so && make -s -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -DPGX_LCE_ENTRY_CAP=16u" libpgx.so)
(cd $D && $B) > gpurun_out/r4_x11_cap16.json 2> gpurun_out/r4_x11_cap16.err; show cap16
$B > gpurun_out/r4_x11_base2.json 2> gpurun_out/r4_x11_base2.err; show base2

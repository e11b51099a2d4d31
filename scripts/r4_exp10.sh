#!/bin/bash
# round 4, experiment 10: where the seed entry of a starting stage waits (LDS through global_load_lds / registers) and who loads the lines of the LCE
# variant (asm statements in place / the compiler), each built in /tmp and run on the default workload
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
run() { # name, extra CXXFLAGS
  D=/tmp/pgx_$1; rm -rf $D; mkdir -p $D
  cp -r pangenome-index_amd include oracle bench.py __graft_entry__.py tests $D/
  (cd $D/pangenome-index_amd && rm -rf build libpgx.so && make -s -j16 CXXFLAGS="-O3 -std=c++17 -fPIC $2" libpgx.so)
  (cd $D && python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-fresh --parity-reads 30000 --steps 10) > gpurun_out/r4_x10_$1.json 2> gpurun_out/r4_x10_$1.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_x10_$1.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]
print("$1: %.1f M reads/s, step %.2f ms, main %.2f ms, lines %.1f M, parity %s" % (d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], r["probes_issued"]/1e6, d["parity_sample"]["identical"]))
PY
}
run lds_asm ""
run reg_asm "-DPGX_SEED_VIA_LDS=0"
run lds_c "-DPGX_LCE_ASM_LOADS=0"
run reg_c "-DPGX_SEED_VIA_LDS=0 -DPGX_LCE_ASM_LOADS=0"

#!/bin/bash
# usage (GPU box): bash scripts/fm_stats.sh [bench args...] -- lane-occupancy diagnostics of pgx_find_mems_kernel
# WLS="chr22 x" selects the workloads (default: x synth)
# (a -DPGX_FM_STATS build of libpgx.so in /tmp; prints wave trips, live lane-trips and the longest wave)
set -e
D=/tmp/pgx_stats_build
if [ ! -f $D/pangenome-index_amd/libpgx.so ] || [ -z "$PGX_STATS_KEEP" ]; then # (PGX_STATS_KEEP=1: reuse the build of an earlier call on this box)
  rm -rf $D; mkdir -p $D
  cp -r pangenome-index_amd include oracle bench.py __graft_entry__.py tests $D/ 2>/dev/null || true
  (cd $D/pangenome-index_amd && rm -rf build libpgx.so && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -DPGX_FM_STATS" libpgx.so)
fi
cd $D
for wl in ${WLS:-x synth}; do
  echo "$wl $*"; PGX_FM_STATS=1 python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-tags --no-fresh --no-parity "$@" 2>&1 | grep -E "pgx\]" | tail -3
done

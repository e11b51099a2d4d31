#!/bin/bash
# usage (GPU box): bash scripts/exp_dup.sh [bench args...]
# Sensitivity builds of the pairs kernel: EXP=PGX_EXP_DUP (default: the popcount section of every trip computed twice, +~110 VALU
# instructions of ~515 per wave trip) or EXP=PGX_EXP_LOAD6 (a sixth 16-byte piece of the probed line), against the ordinary build, same bench run.
set -e
D=/tmp/pgx_dup_build; rm -rf $D; mkdir -p $D
cp -r pangenome-index_amd include oracle bench.py __graft_entry__.py tests $D/ 2>/dev/null || true
cd $D/pangenome-index_amd && rm -rf build libpgx.so && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -D${EXP:-PGX_EXP_DUP}" libpgx.so
cd $D
python3 bench.py --workload chr22 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('${EXP:-PGX_EXP_DUP} build:', d['kernel_ms_per_step'])"

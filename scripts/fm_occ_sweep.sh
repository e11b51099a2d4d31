#!/bin/bash
# usage (GPU box): bash scripts/fm_occ_sweep.sh -- resident workgroups per CU vs find_mems throughput (PGX_FM_WG_PER_CU)
run() { python3 bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), 'Mreads/s', {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"; }
for w in 2 3 4 5 6; do for wl in x synth; do echo "$wl wg/cu=$w"; PGX_FM_WG_PER_CU=$w run --workload $wl; done; done

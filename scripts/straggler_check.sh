#!/bin/bash
# usage (GPU box): bash scripts/straggler_check.sh -- a batch that contains a read from a sequence end (2.9 M synthetic reads,
# seed 42): find_mems kernel time with the heavy-read path (default) and without it (PGX_FM_HEAVY_EXT=0)
run() { python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-tags 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), 'Mreads/s  Gext/s', round(d['extensions_per_s']/1e9,2), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items() if v})"; }
echo "heavy-read path on"; run --workload synth --reads 2900000
echo "heavy-read path off"; PGX_FM_HEAVY_EXT=0 run --workload synth --reads 2900000

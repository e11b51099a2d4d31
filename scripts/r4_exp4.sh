#!/bin/bash
# round 4, experiment 4: run continuation + tag-stage changes: tests, haplotype sweep again, N-read share again
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pairs.py::test_wide_intervals_and_the_run_continuation tests/test_wide_image.py tests/test_gpu_tags_large.py tests/test_gpu_parity.py::test_tag_queries_all_sort_paths tests/test_gpu_fullsize.py::test_synth_pangenome_one_million_reads tests/test_gpu_spec.py -m gpu -x -q --durations=6 > gpurun_out/r4_t5.log 2>&1 || { tail -30 gpurun_out/r4_t5.log; exit 1; }
tail -12 gpurun_out/r4_t5.log
HAPS_EXTRA="--no-cpu-baseline" bash scripts/haps_sweep.sh 64 96 2>&1 | grep -E "^haps|rror" || true
W=/tmp/pgxwd; mkdir -p $W
for f in 0 0.02 0.05; do
  python bench.py --workdir $W --n-read-frac $f --no-cpu-baseline --no-secondary --no-parity --no-fresh --steps 10 > gpurun_out/r4_nfrac_$f.json 2> gpurun_out/r4_nfrac_$f.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_nfrac_$f.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("n-read share $f: step %.2f ms, find_mems %.2f (main %.2f), compact %.2f, locate %.2f, gather %.2f, sort %.2f, %.1f M reads/s, positions %d" % (d["ms_per_step"], k["find_mems"], k["find_mems_main"], k["compact"], k["tag_locate"], k["tag_gather"], k["tag_sort"], d["value"]/1e6, d["positions_per_step"]))
PY
done

#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/gpu_check.sh [tag]
# runs the GPU parity tests, then both bench workloads; prints a short summary
tag=${1:-run}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -4 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 10 --warmup 2 ${BENCH_FLAGS:---no-cpu-baseline} > gpurun_out/${tag}_x.json 2> gpurun_out/${tag}_x.err || { tail -5 gpurun_out/${tag}_x.err; exit 1; }
python bench.py --workload synth --steps 5 --warmup 1 ${BENCH_FLAGS:---no-cpu-baseline} > gpurun_out/${tag}_synth.json 2> gpurun_out/${tag}_synth.err || { tail -5 gpurun_out/${tag}_synth.err; exit 1; }
python3 - <<PY
import json
for w in ("x", "synth"):
    d = json.load(open("gpurun_out/${tag}_%s.json" % w))
    k = {a: round(b, 3) for a, b in d["kernel_ms_per_step"].items()}
    print(w, round(d["value"] / 1e6, 2), "Mreads/s", k, "GB/s", round(d["roofline"]["achieved"]), "frac", round(d["roofline"]["frac"], 3),
          "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY

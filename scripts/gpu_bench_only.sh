#!/bin/bash
tag=${1:-run}
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_x.json 2> gpurun_out/${tag}_x.err || { tail -5 gpurun_out/${tag}_x.err; exit 1; }
python bench.py --workload synth --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_synth.json 2> gpurun_out/${tag}_synth.err || { tail -5 gpurun_out/${tag}_synth.err; exit 1; }
python3 - <<PY
import json
for w in ("x", "synth"):
    d = json.load(open("gpurun_out/${tag}_%s.json" % w))
    k = {a: round(b, 3) for a, b in d["kernel_ms_per_step"].items()}
    print(w, round(d["value"] / 1e6, 2), "Mreads/s", k, "GB/s", round(d["roofline"]["achieved"]))
PY

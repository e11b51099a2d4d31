#!/bin/bash
# round 4, experiment 14 (final kernel): the share of reads over N runs again (0 / 0.4 % default / 2 % / 5 %), and reads/s against the batch size
set -e
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
for f in 0 0.004 0.02 0.05; do
  python bench.py --workdir $W --n-read-frac $f --no-cpu-baseline --no-secondary --no-parity --no-fresh --steps 10 > gpurun_out/r4_x14_nfrac_$f.json 2> gpurun_out/r4_x14_nfrac_$f.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_x14_nfrac_$f.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("n-read share $f: step %.2f ms, find_mems %.2f (main %.2f), compact %.2f, locate %.2f, gather %.2f, sort %.2f, %.1f M reads/s, positions %d" % (d["ms_per_step"], k["find_mems"], k["find_mems_main"], k["compact"], k["tag_locate"], k["tag_gather"], k["tag_sort"], d["value"]/1e6, d["positions_per_step"]))
PY
done
python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-parity --batch-sweep --steps 10 > gpurun_out/r4_x14_sweep.json 2> gpurun_out/r4_x14_sweep.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r4_x14_sweep.json").read().strip().splitlines()[-1])
for r in d.get("batch_size_sweep", []):
    print(r)
PY

#!/bin/bash
# round 4, final measurements: the driver's bench command, rocprofv3 kernel stats + PMC traffic, the n = 4.35e9 workload
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err || { tail -5 gpurun_out/r4_final_bench.err; echo BENCH FAILED; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_final_bench.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]; f=d["fresh_batch"]
print("default: %.1f M reads/s, step %.2f ms, main %.2f, frac %.3f, lines %.1f M + %.1f M seeds, traffic/model %s, parity %s; fresh packed %.1f M/s (%.2f ms, per upload %.2f ms), bytes %.1f M/s; cpu %.0f reads/s on %d cores; x: %.1f M reads/s"
      % (d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], r["frac"], r["probes_issued"]/1e6, r["seed_loads"]/1e6, r.get("traffic_over_model"), d["parity_sample"]["identical"], f["packed"]["value"]/1e6, f["packed"]["ms_per_step"],
         f["packed"]["per_upload_ms"], f["bytes"]["value"]/1e6, d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["secondary"]["value"]/1e6))
PY
bash scripts/profile_round.sh r04 chr22 x 2>&1 | tail -16
python bench.py --workload wg > gpurun_out/r4_final_wg.json 2> gpurun_out/r4_final_wg.err || { tail -5 gpurun_out/r4_final_wg.err; echo WG FAILED; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_final_wg.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; r=d["roofline"]
print("wg: n %d, %.1f M reads/s, step %.2f ms, main %.2f, frac %.3f, lines %.1f M, parity %s, fresh %.1f M/s" % (d["config"]["bwt_size"], d["value"]/1e6, d["ms_per_step"], k["find_mems_main"], r["frac"], r["probes_issued"]/1e6, d["parity_sample"]["identical"], d["fresh_batch"]["packed"]["value"]/1e6))
PY

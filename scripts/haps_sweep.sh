#!/bin/bash
# usage (GPU box): bash scripts/haps_sweep.sh [H ...]   (default 32 64 96) -- VERDICT r03 item 6: the chr22 workload at fixed n ~ 640 M with H haplotypes
# (intervals ~H wide): reads/s, lines per extension, share of lane trips that wait for a second block (diagnostics build), tag stage, parity sample
set -e
mkdir -p gpurun_out
for H in ${@:-32 64 96}; do
  B=$((320000000 / H)); W=/tmp/pgxwd_h$H; mkdir -p $W
  python bench.py --workdir $W --haps $H --base-len $B --no-secondary --no-fresh --no-overlap --cpu-seconds 4 --steps 10 $HAPS_EXTRA > gpurun_out/r4_haps_$H.json 2> gpurun_out/r4_haps_$H.err || { tail -5 gpurun_out/r4_haps_$H.err; exit 1; }
  WLS=chr22 PGX_STATS_KEEP=1 bash scripts/fm_stats.sh --workdir $W --haps $H --base-len $B > gpurun_out/r4_haps_${H}_stats.txt 2>&1 || true
  python - <<PY
import json, re
d = json.loads(open("gpurun_out/r4_haps_$H.json").read().strip().splitlines()[-1])
r, k = d["roofline"], d["kernel_ms_per_step"]
st = open("gpurun_out/r4_haps_${H}_stats.txt").read()
m = re.search(r"live lane-trips (\d+) .*?with two extensions (\d+), waiting for a second block (\d+)", st)
wait = ("%.1f %%" % (100.0 * int(m.group(3)) / int(m.group(1)))) if m else "?"
print("haps $H: n %d, pairs_stride %s, %.1f M reads/s, step %.2f ms, main kernel %.2f ms, frac %.3f, lines/extension %.3f, second-block trips %s, tag stage %.2f ms, MEMs/read %.2f, positions/read %.2f, parity %s"
      % (d["config"]["bwt_size"], d["config"]["pairs_stride"], d["value"] / 1e6, d["ms_per_step"], k["find_mems_main"], r["frac"], 1.0 / r["extensions_per_line"], wait,
         k["tag_locate"] + k["tag_gather"] + k["tag_sort"], d["mems_per_step"] / d["config"]["reads_per_gpu"], d["positions_per_step"] / d["config"]["reads_per_gpu"], d["parity_sample"]["identical"]))
PY
done

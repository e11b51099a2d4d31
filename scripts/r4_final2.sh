#!/bin/bash
# round 4, last measurements of the final kernel: the driver's bench command, rocprofv3 kernel stats + PMC traffic, counters of the dominant kernel, lane
# statistics (diagnostics build), the n = 4.35e9 workload, 32 / 64 / 96 haplotypes at n = 640 M
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/r4_final.sh 2>&1 | grep -E "^default|^wg|GB per launch|FAIL" || true
bash scripts/pmc_pass.sh r04f chr22 SQ_INSTS_VALU VALUBusy SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD TA_BUSY_avr GRBM_GUI_ACTIVE > gpurun_out/r4_final_counters_raw.txt 2>&1 || true
grep -E "pairs_kernel" gpurun_out/r4_final_counters_raw.txt | cut -c1-160 || true
WLS=chr22 bash scripts/fm_stats.sh --workdir /tmp/wd 2>&1 | tail -3 > gpurun_out/r4_final_lane_stats.txt || true
cat gpurun_out/r4_final_lane_stats.txt
for H in 32 64 96; do
  WH=/tmp/pgxwd_h$H; mkdir -p $WH
  python bench.py --workdir $WH --haps $H --base-len $((320000000 / H)) --no-secondary --no-fresh --cpu-seconds 4 --steps 10 > gpurun_out/r4_final_haps_$H.json 2> gpurun_out/r4_final_haps_$H.err || { tail -3 gpurun_out/r4_final_haps_$H.err; continue; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4_final_haps_$H.json").read().strip().splitlines()[-1])
r, k = d["roofline"], d["kernel_ms_per_step"]
print("haps $H: n %d, %.1f M reads/s, step %.2f ms, main kernel %.2f ms, lines %.1f M + %.1f M seed entries, tag stage %.2f ms, MEMs/read %.2f, positions/read %.2f, parity %s"
      % (d["config"]["bwt_size"], d["value"] / 1e6, d["ms_per_step"], k["find_mems_main"], r["probes_issued"] / 1e6, r["seed_loads"] / 1e6,
         k["tag_locate"] + k["tag_gather"] + k["tag_sort"], d["mems_per_step"] / d["config"]["reads_per_gpu"], d["positions_per_step"] / d["config"]["reads_per_gpu"], d["parity_sample"]["identical"]))
PY
done

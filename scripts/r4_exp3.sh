#!/bin/bash
# round 4, experiment 3: batch-size sweep, counters of the LDS-image (x) kernel, kernel breakdown of a batch with 5 % reads over N runs
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
W=/tmp/pgxwd; mkdir -p $W
python bench.py --workdir $W --no-cpu-baseline --no-secondary --no-parity --no-fresh --batch-sweep --steps 10 > gpurun_out/r4_sweep.json 2> gpurun_out/r4_sweep.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_sweep.json").read().strip().splitlines()[-1])
for r in d["batch_size_sweep"]:
    print("batch %8d reads: resident %.1f M/s (%.3f ms, main %.3f), fresh %.1f M/s (%.3f ms)" % (r["reads"], r["resident_reads_per_s"]/1e6, r["resident_ms_per_step"], r["find_mems_main_ms"], r["fresh_reads_per_s"]/1e6, r["fresh_ms_per_step"]))
PY
bash scripts/pmc_pass.sh r4x x VALUBusy LDSBankConflict SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r4_x_counters.txt || true
cat gpurun_out/r4_x_counters.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_n05 -- python3 bench.py --workdir $W --n-read-frac 0.05 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity --no-fresh > gpurun_out/r4_n05.json 2> gpurun_out/r4_n05.err || echo FAIL n05
cat gpurun_out/r4_n05/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-50,150- | head -30

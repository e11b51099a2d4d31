#!/bin/bash
# usage: bash scripts/pmc_sq.sh <tag> <workload> -- SQ instruction-mix counters for the find_mems kernel (separate passes)
tag=$1; wl=${2:-synth}
export TMPDIR=/tmp
R=$PWD/gpurun_out/pmcsq_$tag; mkdir -p $R; W=/tmp/pgxwd_$tag
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); D=$R/p$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --workdir $W > $D.json 2> $D.err || echo FAIL $C
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/p*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-34:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "find_mems" in k:
            print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY

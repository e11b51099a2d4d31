#!/bin/bash
# usage (GPU box): bash scripts/profile_round.sh <tag> [workloads...]   (default: chr22 x)
# For each workload: rocprofv3 --kernel-trace --stats of bench.py (kernel durations), then one --pmc pass per counter
# (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) -> traffic_<workload>.json, which records what it measured
# (bwt_size, reads, min_len, image kind, tags) so that bench.py attaches it only to the same workload.
tag=$1; shift
WLS=${@:-chr22 x}
export TMPDIR=/tmp
R=$PWD/gpurun_out/prof_$tag; mkdir -p $R
for wl in $WLS; do
  steps=10; [ $wl = chr22 ] && steps=5
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/${wl}_stats -- python3 bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-secondary --no-parity --no-fresh --workdir /tmp/wd > $R/${wl}_bench_under_rocprof.json 2> $R/${wl}_stats.err || echo FAIL stats $wl
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/${wl}_$C -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-parity --no-fresh --workdir /tmp/wd > $R/${wl}_$C.json 2> $R/${wl}_$C.err || echo FAIL $C $wl
  done
done
python3 - $R $WLS <<'PY'
import csv, glob, json, collections, sys
R, wls = sys.argv[1], sys.argv[2:]
for wl in wls:
    out = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for f in glob.glob("%s/%s_%s/*/*_counter_collection.csv" % (R, wl, C)):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        # per step (= per pgx_batch_run): a kernel launched several times per step (the dense2 kernel serves the non-ACGT reads on the second
        # stream AND the reads the pairs kernel hands on) counts with all its launches; steps = launches of the compaction kernel
        steps = max(1, len(agg.get("pgx_compact_mems_kernel", [])))
        out[C] = {k: sum(v) / steps for k, v in agg.items() if k.startswith("pgx_")}
    fms = [k for k in out["FETCH_SIZE"] if "find_mems_kernel" in k or "find_mems_pairs_kernel" in k]  # the pairs kernel and the kernel that serves what it hands on
    fm = " + ".join(sorted(fms))
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        out[C][fm] = sum(out[C].get(k, 0.0) for k in fms)
    bench = json.load(open("%s/%s_FETCH_SIZE.json" % (R, wl)))
    cfg = bench["config"]
    # FETCH_SIZE / WRITE_SIZE are in units of 1024 B (rocprofv3).  FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but every read request of
    # this kernel moves a whole 128-byte line (MI355X_MICROARCH.md "HBM": FETCH_SIZE tallies 128-B requests at 64 B; confirmed
    # for THIS access pattern by profiles/r01_ubench_random_gather.txt: random 64-B and 128-B records are served at the same
    # record rate), so the read side is doubled.  WRITE_SIZE is exact.
    rec = {"workload": wl, "recorded": "round 4 (" + sys.argv[1].rsplit("prof_", 1)[-1] + ")", "kernel": fm, "bwt_size": cfg["bwt_size"], "reads": cfg["reads_per_gpu"], "min_len": cfg["min_len"], "tags": cfg["tags"],
           "image_kind": cfg["image_kind"], "image_pairs": cfg.get("image_pairs", 0), "pairs_stride": cfg.get("pairs_stride", 0),
           "rank_image": cfg["rank_image"],
           "FETCH_SIZE_KB_per_launch": out["FETCH_SIZE"][fm], "WRITE_SIZE_KB_per_launch": out["WRITE_SIZE"].get(fm, 0.0),  # per step of 1 batch
           "find_mems_hbm_bytes_per_launch": (2.0 * out["FETCH_SIZE"][fm] + out["WRITE_SIZE"].get(fm, 0.0)) * 1024.0,
           "all_kernels_FETCH_KB": out["FETCH_SIZE"], "all_kernels_WRITE_KB": out["WRITE_SIZE"],
           "note": "memory-side (fabric) bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: Infinity-Cache hits are included, so this is an upper bound on HBM bytes"}
    json.dump(rec, open("%s/traffic_%s.json" % (R, wl), "w"), indent=1)
    print(wl, fm, rec["find_mems_hbm_bytes_per_launch"] / 1e9, "GB per launch")
PY
for wl in $WLS; do cat $R/${wl}_stats/*/*_kernel_stats.csv | head -6 | cut -c1-60,300-; done

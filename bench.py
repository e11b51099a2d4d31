#!/usr/bin/env python3
"""bench.py -- find_mems reads/s on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the whole find_mems path (find_all_mems + tag queries for every read,
src/find_mems.cpp:94-139 without the printing) over one device-resident batch of synthetic
150-bp reads.  Reads shard by rank with the index replicated and no data-path collective
(SURVEY 8e), so scaling is "weak": every rank owns `--reads` reads.

Workloads (config.workload):
  x      BASELINE configs[1]: index built from test_data/x.rl_bwt (committed as tests/golden/x.rl_bwt),
         1M synthetic reads per GPU.  min_len defaults to 10: at the README's other example (20) the
         reference's rank-cache quirk on this no-N index finds zero MEMs (DESIGN.md "Workloads").
  synth  sigma=6 synthetic pangenome (SURVEY 8d config-3 recipe scaled by --base-len): proper FMD
         index where COMPAT == STRICT; BWT built in-process by SA-IS.

Prints ONE JSON line on rank 0.  The CPU oracle is used here only for the `cpu_baseline` leg.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="x", choices=["x", "synth", "chrom"])
    ap.add_argument("--chroms", type=int, default=6, help="chrom: number of synthetic chromosomes (sharded over the ranks)")
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--min-len", type=int, default=None)
    ap.add_argument("--min-occ", type=int, default=1)
    ap.add_argument("--base-len", type=int, default=4_000_000, help="synth: base sequence length")
    ap.add_argument("--haps", type=int, default=8, help="synth: haplotypes (each in both strands)")
    ap.add_argument("--mode", default="compat", choices=["compat", "strict"])
    ap.add_argument("--no-tags", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target core-seconds of the CPU sample")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--pcie", action="store_true",
                    help="also time the host-buffer API (upload + run + download per call); reported separately, never as value")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device rehearses the N>1 path on a single-GPU box")
    ap.add_argument("--one-device", action="store_true", help="every rank uses device 0 (rehearsal only)")
    return ap.parse_args()


def make_workload(args, rank, world, wd, barrier):
    import pgx_workload as W

    golden = os.path.join(ROOT, "tests", "golden")
    if args.workload == "x":
        name = "x"
        if rank == 0:
            W.build_index_from_rlbwt(os.path.join(golden, "x.rl_bwt"), wd, name)
        barrier()
        ri, tags = os.path.join(wd, name + ".ri"), os.path.join(wd, name + ".compact.tags")
        seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
        seed = 42 + 2
        desc = "x.rl_bwt index (n=3012, sigma=5), BASELINE configs[1]"
    else:
        name = "synth_%d_%d" % (args.base_len, args.haps)
        text = os.path.join(wd, name + ".txt")
        if rank == 0 and not os.path.exists(os.path.join(wd, name + ".ri")):
            t0 = time.time()
            W.synth_pangenome_text(text, base_len=args.base_len, n_hap=args.haps, seed=45)
            W.build_index_from_text(text, wd, name)
            sys.stderr.write("[bench] synthetic index built in %.1f s\n" % (time.time() - t0))
        barrier()
        ri, tags = os.path.join(wd, name + ".ri"), os.path.join(wd, name + ".compact.tags")
        seqs = W.load_sequences(text)
        seed = 42 + 3
        desc = "synthetic pangenome: %d bp base x %d haplotypes x 2 strands, sigma=6" % (args.base_len, args.haps)
    cat, offs = W.sample_reads(seqs, args.reads, args.read_len, seed=seed + 1000 * rank)
    return ri, tags, cat, offs, desc


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.min_len is None:
        args.min_len = 10 if args.workload == "x" else 20

    import torch
    import torch.distributed as dist

    import pgx_ffi as P

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libpgx has no CPU fallback")
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"

    def barrier():
        if world > 1:
            dist.barrier()

    own_tmp = None
    if args.workdir:
        wd = args.workdir
        os.makedirs(wd, exist_ok=True)
    else:
        # all ranks of one node share the directory rank 0 fills
        wd = os.path.join(tempfile.gettempdir(), "pgx_bench_%s" % os.environ.get("MASTER_PORT", str(os.getppid())))
        os.makedirs(wd, exist_ok=True)
        own_tmp = wd

    if args.workload == "chrom":
        return run_chrom(args, rank, world, local, wd, barrier, dist, red_dev)
    ri, tags, cat, offs, desc = make_workload(args, rank, world, wd, barrier)
    mode = P.MODE_COMPAT if args.mode == "compat" else P.MODE_STRICT
    idx = P.Index(ri, None if args.no_tags else tags, mode=mode)
    info = idx.info()
    idx.to_device(local)
    batch = idx.batch(cat, offs, device=local)  # inputs resident in HBM before the timed region
    flags = P.RUN_TIMING | (0 if args.no_tags else P.RUN_TAGS)
    stream = torch.cuda.current_stream().cuda_stream

    for _ in range(args.warmup):
        batch.run(args.min_len, args.min_occ, flags, stream)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    k_ms = dict(find_mems=0.0, compact=0.0, tag_locate=0.0, tag_gather=0.0, tag_sort=0.0, total=0.0)
    for _ in range(args.steps):
        batch.run(args.min_len, args.min_occ, flags, stream)
        t = batch.timing()
        for key in k_ms:
            k_ms[key] += getattr(t, "ms_" + key)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_mems, n_pos, n_ext = batch.counts()
    tot = torch.tensor([float(n_mems), float(n_pos), float(n_ext)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    tot = [float(v) for v in tot.tolist()]

    if rank == 0:
        K, n = args.steps, args.reads
        reads_total = n * world * K
        # roofline of the dominant kernel (pgx_find_mems_kernel), SURVEY 8d "Algorithmic bytes":
        #   per extension 2 rank probes x (B_blk + 16 B directory), per read L+1 input bytes,
        #   per MEM 32 output bytes; E_read and MEMs are the kernel's own exact counters.
        b_blk = float(info.ref_block_mean_bytes)
        read_bytes = float(len(cat) + n)
        algo_bytes = n_ext * 2.0 * (b_blk + 16.0) + read_bytes + 32.0 * n_mems
        fm_ms = k_ms["find_mems"] / K
        achieved = algo_bytes / (fm_ms * 1e-3) / 1e9 if fm_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("find_mems_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "find_mems reads/sec (150 bp batch)",
            "value": reads_total / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": ("x.rl_bwt index, %d synthetic %d-bp reads per GPU" % (n, args.read_len)) if args.workload == "x"
                else desc + ", %d reads per GPU" % n,
                "index": desc, "reads_per_gpu": n, "read_len": args.read_len, "min_len": args.min_len,
                "min_occ": args.min_occ, "mode": args.mode, "tags": not args.no_tags,
                "sharding": "reads sharded by rank, index replicated, no collective",
                "bwt_size": int(info.bwt_size), "bwt_runs": int(info.n_runs), "image_in_lds": bool(info.image_in_lds),
            },
            "mems_per_s": tot[0] * K / dt,
            "extensions_per_s": tot[2] * K / dt,
            "mems_per_step": tot[0], "positions_per_step": tot[1], "extensions_per_step": tot[2],
            "kernel_ms_per_step": {k: v / K for k, v in k_ms.items()},
            "roofline": {
                "bound": "hbm", "kernel": "pgx_find_mems_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms": fm_ms, "bytes_per_extension": 2.0 * (b_blk + 16.0),
            },
        }
        if args.pcie:
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                res = idx.find_mems(cat, offs, args.min_len, args.min_occ, tags=not args.no_tags, device=local)
            line["pcie_inclusive_reads_per_s"] = n * reps / (time.perf_counter() - t1)
            line["pcie_inclusive_note"] = "pgx_find_mems_batch: H2D of reads+offsets, all kernels, D2H of MEMs/positions, host CSR copies"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, ri, tags, cat, offs)
        print(json.dumps(line), flush=True)
    batch.free()
    idx.close()
    barrier()
    if world > 1:
        dist.destroy_process_group()
    if own_tmp and rank == 0:
        import shutil

        shutil.rmtree(own_tmp, ignore_errors=True)


def run_chrom(args, rank, world, local, wd, barrier, dist, red_dev):
    """BASELINE configs[4] shape: chromosome-sharded indexes, every read searched in every shard, per-read MEM
    lists exchanged with all_gather (RCCL over xGMI under the nccl backend).  Strong in the index, every rank
    sees all reads: `value` counts reads fully processed against all shards."""
    import torch

    import pgx_ffi as P
    import pgx_shard as S
    import pgx_workload as W

    K = args.chroms
    lengths = [max(200_000, args.base_len // (c + 2)) for c in range(K)]
    owners = S.lpt_assign(lengths, world)
    texts = [os.path.join(wd, "chrom_%d_%d.txt" % (c, lengths[c])) for c in range(K)]
    for c in owners[rank]:  # every rank builds the shards it owns
        if not os.path.exists(os.path.join(wd, "chrom_%d.ri" % c)):
            W.synth_pangenome_text(texts[c], base_len=lengths[c], n_hap=args.haps, seed=100 + c, n_runs=1, n_run_len=(100, 2000))
            W.build_index_from_text(texts[c], wd, "chrom_%d" % c, with_tags=False)
    barrier()
    seqs = []
    for c in range(K):
        seqs += W.load_sequences(texts[c])
    cat, offs = W.sample_reads(seqs, args.reads, args.read_len, seed=42 + 5)  # the same reads on every rank
    idx = {c: P.Index(os.path.join(wd, "chrom_%d.ri" % c)) for c in owners[rank]}
    batches = {c: idx[c].batch(cat, offs, device=local) for c in owners[rank]}
    stream = torch.cuda.current_stream().cuda_stream
    dev = "cuda" if red_dev == "cuda" else "cpu"

    def step():
        parts = {}
        for c in owners[rank]:
            batches[c].run(args.min_len, args.min_occ, 0, stream)
            parts[c] = batches[c].device_result() if dev == "cuda" else batches[c].result()  # nccl: records stay in HBM
        return S.exchange_mems(parts, args.reads, K, dist=dist if world > 1 else None, device=dev)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mo, mems, shard = step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({
            "metric": "find_mems reads/sec (150 bp batch)", "value": args.reads * args.steps / dt, "unit": "reads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "chromosome-sharded: %d synthetic chromosomes over %d ranks, %d reads searched in every shard, "
                                   "all_gather of per-read MEM lists" % (K, world, args.reads),
                       "chrom_lengths": lengths, "owners": owners, "min_len": args.min_len, "min_occ": args.min_occ,
                       "exchange": "torch.distributed all_gather (%s), host-staged records" % args.dist_backend},
            "mems_per_step": int(mo[-1]), "mems_per_s": float(mo[-1]) * args.steps / dt,
        }), flush=True)
    for b in batches.values():
        b.free()
    barrier()
    if world > 1:
        dist.destroy_process_group()


def host_cores(omp_max):
    """threads the CPU leg may really use: affinity mask, capped by the cgroup CPU quota of the box"""
    n = omp_max
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(round(q / per))))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline(args, ri, tags, cat, offs):
    """The oracle (kind "port": the reference cannot be built here) on a bounded sample of the same
    reads, all host cores via OpenMP over reads; timed regions = find_all_mems + tag queries."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O

    r = O.RIndex(ri)
    t = None if args.no_tags else O.Tags(tags, O.TAGS_COMPACT)
    mode = O.MODE_COMPAT if args.mode == "compat" else O.MODE_STRICT
    cores = host_cores(O.lib().orc_max_threads())
    L = args.read_len
    # one thread first (the reference's own execution model): calibrates the per-core rate
    one = min(args.reads, 5000)
    res1 = O.find_mems_batch(r, t, cat[: one * L], offs[: one + 1], args.min_len, args.min_occ, mode=mode, threads=1)
    sec1 = max(res1["seconds_mems"] + res1["seconds_tags"], 1e-6)
    # all cores on a sample worth ~cpu_seconds core-seconds (bounded by the batch), best of 3
    sample = int(min(args.reads, max(one, (one / sec1) * args.cpu_seconds)))
    best = None
    for _ in range(3):
        res = O.find_mems_batch(r, t, cat[: sample * L], offs[: sample + 1], args.min_len, args.min_occ, mode=mode, threads=cores)
        sec = res["seconds_mems"] + res["seconds_tags"]
        best = sec if best is None else min(best, sec)
    return {"value": sample / best, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d reads of the same batch (~%.0f core-seconds), OpenMP schedule(dynamic,256) over reads, "
                      "find_all_mems + tag-query compute only, best of 3 (%.3f s wall)" % (sample, sample / (one / sec1), best),
            "single_thread_reads_per_s": one / sec1, "single_thread_sample": one}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- find_mems reads/s on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the whole find_mems path (find_all_mems + tag queries for every read,
src/find_mems.cpp:94-139 without the printing) over one device-resident batch of synthetic
150-bp reads.  Reads shard by rank with the index replicated and no data-path collective
(SURVEY 8e), so scaling is "weak": every rank owns `--reads` reads.

`python bench.py --gpus N` starts N ranks itself (one child process per GPU, before anything touches a
GPU in this process) unless it already runs under a launcher (WORLD_SIZE set, e.g. torch.distributed.run).

Workloads (config.workload):
  chr22  (default) BASELINE configs[2] / SURVEY 8d config 3 at full scale: synthetic sigma=6 pangenome, 40 Mbp base x 8
         haplotypes x 2 strands (n = 640 M, r = 65 M runs, COMPAT == STRICT), rank image resident in HBM,
         10 M reads per GPU, min_len 20.  The index is built in-process on the host first (SA-IS; minutes).
  synth  the same recipe at --base-len (default 4 Mbp: n = 64 M, Infinity-Cache resident), 1 M reads.
  x      BASELINE configs[1]: index built from test_data/x.rl_bwt (committed as tests/golden/x.rl_bwt), 1 M reads;
         the whole rank image sits in LDS.  min_len defaults to 10: at 20 the reference's rank-cache quirk on this
         no-N index finds zero MEMs (DESIGN.md "Workloads").  Also reported inside the default run as `secondary`.
  chrom  BASELINE configs[4] shape: chromosome-sharded indexes + exchange of per-read MEM lists.
  wg     one merged index over --chroms (default 8) independent synthetic chromosomes, --haps 32 haplotypes each: n = 4.35e9 > 2^32, the
         64-bit (WIDE) images and kernels; built by pgx_build_index_from_texts (per-chromosome suffix arrays + k-way merge).

Prints ONE JSON line on rank 0.  The CPU oracle is used here only for the `cpu_baseline` legs.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GATHER_CEILING_LINES_PER_S = 51.7e9  # dependent random 128-byte lines from a table beyond the memory-side cache (profiles/r01_ubench_random_gather.txt)

DEFAULTS = {  # workload -> (reads per GPU, min_len, base_len)
    "chr22": (10_000_000, 20, 40_000_000),
    "synth": (1_000_000, 20, 4_000_000),
    "x": (1_000_000, 10, None),
    "chrom": (1_000_000, 20, 4_000_000),
    # whole-genome scale: --chroms independent chromosomes in ONE index of more than 2^32 symbols (8 x 8.5 Mbp x 32 haplotypes x 2 strands = 4.35e9)
    "wg": (10_000_000, 20, 8_500_000),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="chr22", choices=sorted(DEFAULTS))
    ap.add_argument("--chroms", type=int, default=None, help="chrom: synthetic chromosomes sharded over the ranks (default 6); wg: chromosomes of the merged index (default 8)")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--min-len", type=int, default=None)
    ap.add_argument("--min-occ", type=int, default=1)
    ap.add_argument("--base-len", type=int, default=None, help="chr22 / synth / chrom: base sequence length")
    ap.add_argument("--haps", type=int, default=None, help="chr22 / synth: haplotypes (each in both strands), default 8; wg: default 32")
    ap.add_argument("--mode", default="compat", choices=["compat", "strict"])
    ap.add_argument("--no-tags", action="store_true")
    ap.add_argument("--n-read-frac", type=float, default=None,
                    help="share of the reads that overlap an N run of the text (default: wherever uniform sampling falls, 0.4 %% on the chr22 workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="chr22: skip the nested x (configs[1]) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target core-seconds of the CPU sample")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the measured batch (parity_sample)")
    ap.add_argument("--no-overlap", action="store_true", help="(kept for old command lines; the measurement is off unless --overlap)")
    ap.add_argument("--overlap", action="store_true", help="extra measurement with two resident batches in flight (two_batches_in_flight); off by default since round 4: "
                                                          "measured three ways it is within +-3 %% of the bench value (profiles/r04_overlap_by_occupancy.txt)")
    ap.add_argument("--no-fresh", action="store_true", help="skip the fresh-batch measurement (fresh_batch: every step uploads a different batch from pinned host memory, runs it and downloads its results)")
    ap.add_argument("--fresh-workers", type=int, default=4, help="fresh_batch: device batches / host threads / streams in flight")
    ap.add_argument("--batch-sweep", action="store_true", help="also report reads/s against the batch size (64 k ... --reads), resident and fresh (batch_size_sweep)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --reads per GPU; strong (BASELINE configs[3] as worded): ONE batch of --reads cut into --gpus contiguous slices, one per rank")
    ap.add_argument("--parity-reads", type=int, default=100_000, help="reads of the measured batch the oracle re-computes (parity_sample)")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--pcie", action="store_true",
                    help="also time the host-buffer API (upload + run + download per call); reported separately, never as value")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device rehearses the N>1 path on a single-GPU box")
    ap.add_argument("--one-device", action="store_true", help="every rank uses device 0 (rehearsal only)")
    ap.add_argument("--stub-workload", action="store_true",
                    help="launcher self-test: no GPU work, every rank contributes a constant (tests/test_bench_launch.py)")
    args = ap.parse_args(argv)
    d = DEFAULTS[args.workload]
    if args.reads is None:
        args.reads = d[0]
    if args.min_len is None:
        args.min_len = d[1]
    if args.base_len is None:
        args.base_len = d[2]
    if args.haps is None:
        args.haps = 32 if args.workload == "wg" else 8
    if args.chroms is None:
        args.chroms = 8 if args.workload == "wg" else 6
    return args


# ----------------------------------------------------------------------------------------------------------------------
# --gpus N without an external launcher: N child processes, one rank each, started before this process touches a GPU
def free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def fan_out(n, argv):
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0 inherits stdout (its one JSON line is the bench's output); the other ranks print nothing on stdout
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("[bench] ranks failed: %s\n" % bad)
        return 1
    return 0


# ----------------------------------------------------------------------------------------------------------------------
def synth_index(W, wd, base_len, haps, rank, barrier):
    """the sigma = 6 synthetic pangenome index of SURVEY 8d config 3 (scaled by base_len); rank 0 builds, all ranks load"""
    name = "synth_%d_%d" % (base_len, haps)
    text = os.path.join(wd, name + ".txt")
    done = os.path.join(wd, name + ".done")
    build_s = 0.0
    if rank == 0 and not os.path.exists(done):
        t0 = time.time()
        W.synth_pangenome_text(text, base_len=base_len, n_hap=haps, seed=45)
        W.build_index_from_text(text, wd, name)
        open(done, "w").write("ok\n")
        build_s = time.time() - t0
        sys.stderr.write("[bench] synthetic index built in %.1f s (host: SA-IS over eight groups of sequences + k-way merge)\n" % build_s)
    barrier()
    return os.path.join(wd, name + ".ri"), os.path.join(wd, name + ".compact.tags"), text, build_s


def make_workload(args, workload, rank, wd, barrier, n_reads, base_len):
    import pgx_workload as W

    golden = os.path.join(ROOT, "tests", "golden")
    build_s = 0.0
    if workload == "x":
        name = "x"
        if rank == 0:
            W.build_index_from_rlbwt(os.path.join(golden, "x.rl_bwt"), wd, name)
        barrier()
        ri, tags = os.path.join(wd, name + ".ri"), os.path.join(wd, name + ".compact.tags")
        seqs = W.load_sequences(os.path.join(golden, "x.newline_separated"))
        seed = 42 + 2
        desc = "x.rl_bwt index (n=3012, sigma=5), BASELINE configs[1]"
    elif workload == "wg":
        name = "wg_%d_%d_%d" % (args.chroms, base_len, args.haps)
        done = os.path.join(wd, name + ".done")
        texts = [os.path.join(wd, "%s_chr%d.txt" % (name, c)) for c in range(args.chroms)]
        if rank == 0 and not os.path.exists(done):
            t0 = time.time()
            # (two short N runs per haplotype: sigma = 6 as everywhere, without making the k-way merge compare through hundreds of kilobases of N)
            W.synth_chromosome_texts(wd, name, args.chroms, base_len, args.haps, seed=45, n_runs=2, n_run_len=(1000, 10000))
            t1 = time.time()
            W.build_index_from_texts(texts, wd, name)
            open(done, "w").write("ok\n")
            build_s = time.time() - t0
            sys.stderr.write("[bench] %d chromosome texts in %.1f s, merged index built in %.1f s (host: SA-IS per chromosome + k-way merge)\n" % (args.chroms, t1 - t0, time.time() - t1))
        barrier()
        ri, tags = os.path.join(wd, name + ".ri"), os.path.join(wd, name + ".compact.tags")
        seqs = []
        for t in texts:
            seqs += W.load_sequences(t)
        seed = 42 + 5
        desc = "synthetic whole-genome-scale pangenome: %d chromosomes x %d bp base x %d haplotypes x 2 strands in one index, sigma=6" % (args.chroms, base_len, args.haps)
    else:
        ri, tags, text, build_s = synth_index(W, wd, base_len, args.haps, rank, barrier)
        seqs = W.load_sequences(text)
        seed = 42 + 3
        desc = "synthetic pangenome: %d bp base x %d haplotypes x 2 strands, sigma=6" % (base_len, args.haps)
        if workload == "chr22":
            desc += " (chr22-scale, SURVEY 8d config 3 = BASELINE configs[2])"
    nf = args.n_read_frac if workload != "x" else None
    strong = getattr(args, "scaling", "weak") == "strong"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # weak: every rank samples its own n_reads; strong (BASELINE configs[3] as worded): ONE batch of n_reads, the same on every rank, cut into
    # contiguous slices (make_slice below)
    cat, offs = W.sample_reads(seqs, n_reads, args.read_len, seed=seed + (0 if strong else 1000 * rank), n_frac=nf)
    if strong and world > 1:
        cat, offs = strong_slice(cat, offs, rank, world)

    def resample(k):  # further batches of the same shape (fresh_batch rotates over three)
        return W.sample_reads(seqs, len(offs) - 1, args.read_len, seed=seed + 1000 * rank + 7919 * k, n_frac=nf)

    make_workload.resample = resample
    return ri, tags, cat, offs, desc, build_s


def strong_slice(cat, offs, rank, world):
    """the contiguous slice of one batch that `rank` of `world` serves (reads [n r / world, n (r + 1) / world)), offsets rebased to 0"""
    n = len(offs) - 1
    lo, hi = n * rank // world, n * (rank + 1) // world
    o = offs[lo:hi + 1]
    return cat[int(o[0]):int(o[-1])], o - o[0]


def measure(P, idx, cat, offs, local, min_len, min_occ, flags, steps, warmup, stream, sync, barrier):
    """W warm-up steps, then exactly K timed steps between barrier + device synchronisation on both sides"""
    batch = idx.batch(cat, offs, device=local)  # inputs resident in HBM before the timed region
    for _ in range(warmup):
        batch.run(min_len, min_occ, flags, stream)
    sync()
    barrier()
    t0 = time.perf_counter()
    k_ms = dict(find_mems=0.0, find_mems_main=0.0, compact=0.0, tag_locate=0.0, tag_gather=0.0, tag_sort=0.0, total=0.0)
    for _ in range(steps):
        batch.run(min_len, min_occ, flags, stream)
        t = batch.timing()  # HIP events recorded on the launch stream inside pgx_batch_run
        for key in k_ms:
            k_ms[key] += getattr(t, "ms_" + key)
    sync()
    barrier()
    dt = time.perf_counter() - t0
    n_mems, n_pos, n_ext = batch.counts()
    return batch, dt, {k: v / steps for k, v in k_ms.items()}, (n_mems, n_pos, n_ext)


def measure_two_in_flight(idx, cat, offs, local, min_len, min_occ, flags, steps, sync):
    """NOT the bench value: the same K steps with TWO resident batches of the same reads in flight, each on its own stream and driven by its own host
    thread, as the CLI's device workers run consecutive batches -- one batch's compaction and tag stage then run under the other's find_mems kernel.
    Every step is still a complete pgx_batch_run of a whole batch."""
    import threading
    bs = [idx.batch(cat, offs, device=local) for _ in range(2)]
    for b in bs:
        for _ in range(2):
            b.run(min_len, min_occ, flags, 0)  # stream 0: the batch's own stream
    sync()
    share = [steps - steps // 2, steps // 2]
    err = []

    def work(b, k):
        try:
            for _ in range(k):
                b.run(min_len, min_occ, flags, 0)
        except Exception as e:  # noqa: BLE001
            err.append(e)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(b, k)) for b, k in zip(bs, share)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sync()
    dt = time.perf_counter() - t0
    counts = [b.counts() for b in bs]
    for b in bs:
        b.free()
    if err:
        raise err[0]
    return dt, counts


class HostBatch:
    """one batch of reads resident in PINNED host memory in both forms pgx takes: bytes + offsets (pgx_batch_upload) and packed words + offsets
    + the listed reads with a byte outside A C G T (pgx_batch_upload_packed; packed by pgx_pack_reads before the timed region, as the CLI's
    parse threads do it while they parse)"""

    def __init__(self, P, cat, offs, keep_bytes=True):
        import numpy as np

        self.n = len(offs) - 1
        self.offs = P.pinned_array(len(offs), np.uint64)
        self.offs[:] = offs
        words = (len(cat) + 15) // 16
        self.packed = P.pinned_array(max(words, 1), np.uint32)
        side_cap_ids, side_cap_bytes = self.n // 8 + 1024, len(cat) // 8 + 4096
        self.side_ids = P.pinned_array(side_cap_ids, np.uint64)
        self.side_bytes = P.pinned_array(side_cap_bytes, np.uint8)
        t0 = time.perf_counter()
        self.n_side, self.n_side_bytes = P.pack_reads(cat, offs, self.packed, self.side_ids, self.side_bytes)
        self.pack_s = time.perf_counter() - t0
        self.cat = None
        if keep_bytes:
            self.cat = P.pinned_array(max(len(cat), 1), np.uint8)
            self.cat[: len(cat)] = cat
        self.h2d_bytes_packed = words * 4 + 8 * (self.n + 1) + 8 * self.n_side + self.n_side_bytes
        self.h2d_bytes_plain = len(cat) + 8 * (self.n + 1)


def measure_fresh(idx, host_batches, local, min_len, min_occ, flags, steps, sync, packed, workers=4, warm=2):
    """NOT the bench value: K steps that each upload a batch DIFFERENT from the one the device batch held (rotating over the pinned host
    batches), run it and download its results into pinned host arrays -- `workers` device batches, a host thread and a stream each, so that the
    upload of one, the kernels of another and the download of a third overlap (what the find_mems CLI's device workers do).  Returns
    (seconds, per-step counts of the last step of every worker)."""
    import threading

    dbs = [idx.batch_empty(device=local) for _ in range(workers)]
    nb = len(host_batches)
    err, last = [], [None] * workers
    phase = [[0.0, 0.0, 0.0] for _ in range(workers)]  # host seconds inside upload / run / result per worker (timed steps only)

    def one(w, k, timed=False):
        hb = host_batches[(w + k) % nb]
        t0 = time.perf_counter()
        if packed:
            dbs[w].upload_packed(hb.packed, hb.offs, hb.side_ids, hb.side_bytes, hb.n_side)
        else:
            dbs[w].upload(hb.cat, hb.offs)
        t1 = time.perf_counter()
        dbs[w].run(min_len, min_occ, flags & ~2, 0)  # (no event timing inside; the batch's own stream)
        t2 = time.perf_counter()
        nm, npos = dbs[w].result_counts()  # D2H of offsets, MEMs, run counts, positions into the batch's pinned arrays
        t3 = time.perf_counter()
        if timed:
            phase[w][0] += t1 - t0; phase[w][1] += t2 - t1; phase[w][2] += t3 - t2
        last[w] = ((w + k) % nb, nm, npos)

    def work(w, k0, k1, timed=False):
        try:
            for k in range(k0, k1):
                one(w, k, timed)
        except Exception as e:  # noqa: BLE001
            err.append(e)

    for w in range(workers):  # warm-up: allocations, shapes for the speculative sizing
        work(w, 0, warm)
    sync()
    if err:
        raise err[0]
    share = [steps // workers + (1 if w < steps % workers else 0) for w in range(workers)]
    th = [threading.Thread(target=work, args=(w, warm, warm + share[w], True)) for w in range(workers)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    sync()
    dt = time.perf_counter() - t0
    if err:
        raise err[0]
    measure_fresh.phase_ms = [1e3 * sum(p[i] for p in phase) / max(steps, 1) for i in range(3)]  # mean host ms per step inside each call
    return dt, last, dbs


def fresh_record(P, idx, host_batches, local, args, flags, sync, resident_result, resident_counts):
    """the fresh_batch object of the bench line (+ per_upload_ms of both upload forms)"""
    import numpy as np

    n, K = host_batches[0].n, args.steps
    out = {"what": "K steps that each pgx_batch_upload[_packed] a batch different from the one before (three batches resident in pinned host memory, rotating), "
                   "pgx_batch_run it and pgx_batch_result it into pinned host arrays; %d device batches / host threads / streams, so H2D, kernels and D2H of "
                   "consecutive steps overlap (chr22 scale, 3 / 4 / 5 workers: 281-288 / 322 / 257 M reads/s); whole batches of %d reads, tags included" % (args.fresh_workers, n),
           "unit": "reads/s", "steps": K, "workers": args.fresh_workers}
    for key, packed in (("packed", True), ("bytes", False)):
        dt, last, dbs = measure_fresh(idx, host_batches, local, args.min_len, args.min_occ, flags, K, sync, packed, workers=args.fresh_workers)
        rec = {"value": n * K / dt, "ms_per_step": 1e3 * dt / K,
               "host_ms_per_step_inside": dict(zip(("upload", "run", "result"), measure_fresh.phase_ms)),
               "h2d_MB_per_step": (host_batches[0].h2d_bytes_packed if packed else host_batches[0].h2d_bytes_plain) / 1e6}
        # a step that uploaded host batch 0 must give exactly what the resident (measured) batch gave: full arrays, device vs device
        ident = None
        for w, l in enumerate(last):
            if l is not None and l[0] == 0 and resident_result is not None:
                r = dbs[w].result()
                ident = bool(np.array_equal(r["mem_offsets"], resident_result["mem_offsets"]) and r["mems"].tobytes() == resident_result["mems"].tobytes()
                             and ("positions" not in resident_result or (np.array_equal(r["pos_offsets"], resident_result["pos_offsets"])
                                                                         and np.array_equal(r["positions"], resident_result["positions"]))))
                del r
                break
        rec["identical_to_the_resident_batch"] = ident
        # per-upload device passes, timed by HIP events on one worker: upload batch 0 again, then the first run after it
        db = dbs[0]
        hb = host_batches[0]
        if packed:
            db.upload_packed(hb.packed, hb.offs, hb.side_ids, hb.side_bytes, hb.n_side)
        else:
            db.upload(hb.cat, hb.offs)
        db.run(args.min_len, args.min_occ, flags | 2, 0)
        rec["per_upload_ms"] = float(db.timing().ms_per_upload)
        rec["counts_equal_the_bench_steps"] = bool(tuple(db.counts()) == tuple(resident_counts))
        for b in dbs:
            b.free()
        out[key] = rec
    out["value"] = out["packed"]["value"]
    out["ms_per_step"] = out["packed"]["ms_per_step"]
    out["per_upload_ms"] = out["packed"]["per_upload_ms"]
    out["pack_on_host_s_per_batch"] = host_batches[0].pack_s
    out["per_upload_ms_what"] = ("device time before the first find_mems launch that a fresh batch pays and a re-run of resident reads (the bench value) does not: "
                                 "packed upload = unpack to bytes + the listed reads' bytes + the MEM slot scan; byte upload = the pass that packs the reads and finds bytes outside "
                                 "A C G T, its read-back, the pass that lists those reads, the slot scan (pgx_timing.ms_per_upload)")
    return out


def batch_size_sweep(P, idx, cat, offs, local, args, flags, stream, sync):
    """reads/s against the batch size: the resident re-run rate (the bench value's definition) and the fresh-batch rate (packed uploads), prefixes
    of the measured batch"""
    L = args.read_len
    rows = []
    size = 65536
    sizes = []
    while size < args.reads:
        sizes.append(size)
        size *= 4
    sizes.append(args.reads)
    for m in sizes:
        c, o = cat[: m * L], offs[: m + 1]
        steps = max(5, min(40, int(40e6 // m)))
        batch, dt, k_ms, counts = measure(P, idx, c, o, local, args.min_len, args.min_occ, flags, steps, 2, stream, sync, lambda: None)
        batch.free()
        hbs = [HostBatch(P, c, o, keep_bytes=False)]
        dtf, _, dbs = measure_fresh(idx, hbs, local, args.min_len, args.min_occ, flags, max(steps, 6), sync, True)
        for b in dbs:
            b.free()
        rows.append({"reads": m, "resident_reads_per_s": m * steps / dt, "resident_ms_per_step": 1e3 * dt / steps, "find_mems_main_ms": k_ms["find_mems_main"],
                     "fresh_reads_per_s": m * max(steps, 6) / dtf, "fresh_ms_per_step": 1e3 * dtf / max(steps, 6)})
        del hbs
    return rows


def roofline_record(info, cat_len, n_reads, counts, k_ms, timing, workload_key, min_len, tags):
    """Roofline of the dominant kernel = the first launch of the find_mems stage (pgx_find_mems_pairs_kernel where the index has a PAIRS
    image, else pgx_find_mems_kernel), timed on its own by HIP events on the launch stream (pgx_timing.ms_find_mems_main).

    frac       bytes the kernel actually moves / its time / the HBM peak.  The bytes are counted by the kernel itself (pgx_timing.main_lines,
               main_seed_loads: lane trips that fetch a 128-byte line of the rank image, seed / end table entries read) x 128 B, plus what
               it streams: the read bytes and offsets once, 32 B per MEM written, 4 B per read of MEM counts.  Always <= 1.
    algorithmic_ratio   the SURVEY 8d figure: bytes the REFERENCE layout would move for the same extensions (2 rank probes x (B_blk + 16 B)
               per extension, L+1 input bytes per read, 32 B per MEM) over the time of the whole find_mems stage, relative to the HBM
               peak.  Seeds and two-step trips answer extensions without the probes this charges for, so it may exceed 1: a speed-up
               over the reference layout at the roofline, not a bandwidth."""
    n_mems, _, n_ext = counts
    b_blk = float(info.ref_block_mean_bytes)
    algo_bytes = n_ext * 2.0 * (b_blk + 16.0) + float(cat_len + n_reads) + 32.0 * n_mems
    fm_ms, main_ms = k_ms["find_mems"], k_ms["find_mems_main"]
    in_lds = bool(info.image_in_lds)
    pairs = bool(getattr(info, "image_pairs", 0)) and bool(timing.pairs_reads)
    lines, seeds = int(timing.main_lines), int(timing.main_seed_loads)
    all_lines = lines + int(timing.other_lines)
    # (a seed entry is 16 bytes of a random line: a 128-byte line from HBM where the table is gigabytes; the 16 MiB table of an LDS-staged image stays in
    #  the caches and the fabric moves 64-byte sectors for it -- PMC record of the x workload: 55 B per entry)
    seed_bytes = 64.0 if in_lds else 128.0
    moved = 128.0 * lines + seed_bytes * seeds + float(cat_len) + 8.0 * (n_reads + 1) + 32.0 * n_mems + 4.0 * n_reads
    achieved = moved / (main_ms * 1e-3) / 1e9 if main_ms > 0 else 0.0
    rec = {
        # with the rank image staged in LDS nothing of it comes from HBM: issue slots / LDS bound the kernel and frac only says how little
        # of the HBM roofline it needs; otherwise the bound is random 128-byte line fetches from HBM (or the memory-side cache)
        "bound": "lds/valu" if in_lds else "hbm",
        "kernel": "pgx_find_mems_pairs_kernel" if pairs else "pgx_find_mems_kernel",
        "kernel_variant": {0: "one extension per trip", 1: "two-step, byte windows", 2: "two-step, packed reads", 3: "two-step, packed reads, cooperative line fetches",
                           4: "two-step, packed reads, narrow forward stages through the suffix array and the text (LCE)"}.get(int(timing.pairs_reads), "?"),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "frac_note": ("bytes moved per launch (kernel-counted image lines and seed entries x 128 B + streamed reads, offsets, MEM slots) / kernel time / HBM peak"
                      + ("; the rank image is staged in LDS, so the kernel is bound by LDS / issue slots, not by HBM" if in_lds else "")),
        "kernel_ms": main_ms, "bytes_moved_per_launch": moved,
        "probes_issued": lines, "seed_loads": seeds, "lines_per_s": (lines + seeds) / (main_ms * 1e-3) if main_ms > 0 else 0.0,
        "frac_of_gather_ceiling": ((lines + seeds) / (main_ms * 1e-3) / GATHER_CEILING_LINES_PER_S) if (main_ms > 0 and not in_lds) else None,
        "gather_ceiling_note": "dependent random 128-byte lines/s of scripts/ubench_gather.hip on a 2 GB table (profiles/r01_ubench_random_gather.txt)",
        "extensions_per_line": (n_ext / all_lines) if all_lines else None,
        "two_step_trips": int(timing.two_step_trips),
        "other_launches": {"lines": int(timing.other_lines), "seed_loads": int(timing.other_seed_loads), "ms": max(fm_ms - main_ms, 0.0),
                           "what": "pgx_find_mems_kernel over the reads handed on / on the second stream, heavy-read kernel"},
        "traffic": None,
        # LDS-staged image: what the kernel is bound by instead (one rocprofv3 --pmc pass per counter on this workload, kept under profiles/)
        "lds_valu_utilisation": ({"VALUBusy_pct": 53.1, "LDSBankConflict_pct": 0.51, "wait_for_LDS_instructions": "negligible (SQ_WAIT_INST_LDS 8e4 of 9.7e8 wave cycles)",
                                  "source": "profiles/r04_x_kernel_counters.txt (round 4, x workload, 1 M reads)"} if in_lds else None),
        "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_ratio": (algo_bytes / (fm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if fm_ms > 0 else None,
        "algorithmic_ratio_note": "SURVEY 8d bytes of the reference layout for the same extensions / time of the find_mems stage / HBM peak; may exceed 1 (work avoided, not bandwidth)",
        "stage_ms": fm_ms, "bytes_per_extension": 2.0 * (b_blk + 16.0), "extensions_per_launch": n_ext,
    }
    # counter traffic only from a PMC record of THIS workload (scripts/profile_round.sh writes what it measured)
    tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload_key)
    if os.path.exists(tfile):
        try:
            t = json.load(open(tfile))
            same = (int(t.get("bwt_size", -1)) == int(info.bwt_size) and int(t.get("reads", -1)) == int(n_reads)
                    and int(t.get("min_len", -1)) == int(min_len) and int(t.get("image_kind", -1)) == int(info.image_kind)
                    and int(t.get("image_pairs", 0)) == int(info.image_pairs) and bool(t.get("tags", True)) == bool(tags)
                    and int(t.get("pairs_stride", 96 if info.image_pairs else 0)) == int(info.pairs_stride))
            if same:
                rec["traffic"] = t.get("find_mems_hbm_bytes_per_launch")
                rec["traffic_source"] = ("profiles/traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, recorded %s; all find_mems launches of a step)"
                                         % (workload_key, t.get("recorded", "in round 2")))
                if rec["traffic"]:
                    total_model = moved + 128.0 * int(timing.other_lines) + seed_bytes * int(timing.other_seed_loads)
                    rec["traffic_over_model"] = rec["traffic"] / total_model
        except Exception:
            pass
    # ... and what the dominant kernel is bound by when it is not the lines: counters of THIS workload's kernel (scripts/r4_final2.sh -> profiles/counters_<workload>.json)
    cfile = os.path.join(ROOT, "profiles", "counters_%s.json" % workload_key)
    if os.path.exists(cfile) and not in_lds:
        try:
            c = json.load(open(cfile))
            if int(c.get("bwt_size", -1)) == int(info.bwt_size) and int(c.get("reads", -1)) == int(n_reads) and c.get("kernel_variant") == rec.get("kernel_variant"):
                rec["lds_valu_utilisation"] = {k: c[k] for k in ("VALUBusy_pct", "valu_instructions_per_launch", "salu_instructions_per_launch", "wave_trips", "valu_per_wave_trip",
                                                                "lane_trips_per_read", "TA_busy_pct", "source") if k in c}
        except Exception:
            pass
    return rec


_ORACLE = {}


def oracle_objects(ri, tags):
    """the oracle's view of the index files (loaded once per bench run; used by the cpu_baseline and parity_sample legs only)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O

    key = (ri, tags)
    if key not in _ORACLE:
        _ORACLE.clear()  # one index at a time (the chr22-scale oracle structures take gigabytes)
        _ORACLE[key] = (O.RIndex(ri), None if tags is None else O.Tags(tags, O.TAGS_COMPACT))
    return _ORACLE[key]


def parity_sample(args, ri, tags, cat, offs, min_len, res, sample, idx=None, local=0):
    """The oracle on the first `sample` reads of the batch against the prefix of the device result of the SAME batch run (one chunk of
    n reads, the measured configuration): MEM offsets, MEM bytes, tag run counts, positions.  The oracle is the checker here, nothing else."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np

    import oracle_ffi as O

    L = args.read_len
    r, t = oracle_objects(ri, tags)
    mode = O.MODE_COMPAT if args.mode == "compat" else O.MODE_STRICT
    cores = host_cores(O.lib().orc_max_threads())
    ref = O.find_mems_batch(r, t, cat[: sample * L], offs[: sample + 1], min_len, args.min_occ, mode=mode, threads=cores)
    nm = int(ref["mem_offsets"][-1])
    ok = bool(np.array_equal(res["mem_offsets"][: sample + 1], ref["mem_offsets"])
              and res["mems"][:nm].tobytes() == ref["mems"].tobytes())
    npos = 0
    if ok and t is not None:
        npos = int(ref["pos_offsets"][-1])
        ok = bool(np.array_equal(res["tag_run_counts"][:nm], ref["tag_run_counts"])
                  and np.array_equal(res["pos_offsets"][: nm + 1], ref["pos_offsets"])
                  and np.array_equal(res["positions"][:npos], ref["positions"]))
    ext_ok = None
    if idx is not None:  # the same reads as a batch of their own: everything again, and the extension counter (a total per run) against the oracle's
        own = idx.find_mems(cat[: sample * L], offs[: sample + 1], min_len, args.min_occ, tags=t is not None, device=local)
        ext_ok = bool(own["n_extensions"] == ref["n_extensions"] and np.array_equal(own["mem_offsets"], ref["mem_offsets"])
                      and own["mems"].tobytes() == ref["mems"].tobytes()
                      and (t is None or (np.array_equal(own["pos_offsets"], ref["pos_offsets"]) and np.array_equal(own["positions"], ref["positions"]))))
        ok = ok and ext_ok
    return {"reads": int(sample), "mems": nm, "positions": npos, "extensions_of_the_sample": int(ref["n_extensions"]), "identical": ok,
            "sample_as_its_own_batch_identical_incl_n_extensions": ext_ok,
            "what": "oracle (CPU restatement of the reference) on the first reads of the measured batch vs the prefix of the device result of the whole batch"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(fan_out(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import datetime

    import torch
    import torch.distributed as dist

    # under a launcher (WORLD_SIZE set) the process group is created even for one rank: the same RCCL barrier / all-reduce code
    # runs at N = 1 as at N = 8
    use_dist = "WORLD_SIZE" in os.environ
    if use_dist and not args.stub_workload and args.dist_backend == "nccl":
        torch.cuda.set_device(0 if args.one_device else local)
    if use_dist:
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=40))
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"

    def barrier():
        if use_dist:
            dist.barrier()

    if args.stub_workload:  # launcher self-test (CPU tier): no GPU, the line still goes through the same reductions
        import numpy as np

        ones = torch.ones(1, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(ones)
        # the reads every rank would serve: its own batch (weak) or its slice of the one batch (strong), with the slicing of the real workloads
        offs = np.arange(args.reads + 1, dtype=np.uint64) * np.uint64(args.read_len)
        cat = np.zeros(args.reads * args.read_len, dtype=np.uint8)
        first = 0
        if args.scaling == "strong" and world > 1:
            first = args.reads * rank // world
            cat, offs = strong_slice(cat, offs, rank, world)
        mine = torch.tensor([float(len(offs) - 1), float(first), float(len(cat))], dtype=torch.float64)
        allr = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
        if use_dist:
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        if rank == 0:
            print(json.dumps({"metric": "stub", "value": float(world), "unit": "ranks", "n_gpus": world, "n_ranks_seen": int(ones.item()),
                              "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling,
                              "reads_per_rank": [int(t[0].item()) for t in allr], "first_read_per_rank": [int(t[1].item()) for t in allr],
                              "read_bytes_per_rank": [int(t[2].item()) for t in allr]}), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    import pgx_ffi as P

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libpgx has no CPU fallback")
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    # RCCL (or gloo in a rehearsal) sees every rank: sum of ones
    ones = torch.ones(1, dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(ones)
    n_ranks_seen = int(ones.item())

    own_tmp = None
    if args.workdir:
        wd = args.workdir
        os.makedirs(wd, exist_ok=True)
    else:
        # all ranks of one node share the directory rank 0 fills
        wd = os.path.join(tempfile.gettempdir(), "pgx_bench_%s" % os.environ.get("MASTER_PORT", str(os.getpid())))
        os.makedirs(wd, exist_ok=True)
        own_tmp = wd

    if args.workload == "chrom":
        return run_chrom(args, rank, world, local, wd, barrier, dist, red_dev)
    t_prep = time.time()
    ri, tags, cat, offs, desc, build_s = make_workload(args, args.workload, rank, wd, barrier, args.reads, args.base_len)
    mode = P.MODE_COMPAT if args.mode == "compat" else P.MODE_STRICT
    idx = P.Index(ri, None if args.no_tags else tags, mode=mode)
    info = idx.info()
    idx.to_device(local)
    prep_s = time.time() - t_prep
    flags = P.RUN_TIMING | (0 if args.no_tags else P.RUN_TAGS)
    stream = torch.cuda.current_stream().cuda_stream

    batch, dt, k_ms, counts = measure(P, idx, cat, offs, local, args.min_len, args.min_occ, flags, args.steps, args.warmup, stream,
                                      torch.cuda.synchronize, barrier)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    tot = torch.tensor([float(c) for c in counts], dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    tot = [float(v) for v in tot.tolist()]

    # reads this step processed on all ranks (weak: n per rank; strong: the slices of one batch of n)
    nloc = torch.tensor([float(len(offs) - 1)], dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(nloc, op=dist.ReduceOp.SUM)
    parity_failed = False
    if rank == 0:
        K, n = args.steps, len(offs) - 1
        reads_total = int(nloc.item()) * K
        kinds = {P.IMAGE_RL: "run-length blocks", P.IMAGE_DENSE: "dense bit planes", P.IMAGE_DENSE2: "dense2 bit planes"}
        image = (("LDS copy of " if info.image_in_lds else "") + kinds[info.image_kind] + (" + two-step pairs image (a 128-byte block of 96 positions every %d)" % info.pairs_stride if info.image_pairs else "")
                 + (" (64-bit form: counts as deltas against superblock bases)" if info.image_wide else ""))
        line = {
            "metric": "find_mems reads/sec (150 bp batch)",
            "value": reads_total / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": {"chr22": "BASELINE configs[2]: chr22-scale synthetic pangenome index (n = %d), %d synthetic %d-bp reads per GPU, 1 MI355X per rank" % (info.bwt_size, n, args.read_len),
                             "synth": desc + ", %d reads per GPU" % n,
                             "wg": desc + " (n = %d), %d reads per GPU" % (info.bwt_size, n),
                             "x": "BASELINE configs[1]: x.rl_bwt index, %d synthetic %d-bp reads per GPU" % (n, args.read_len)}[args.workload],
                "index": desc, "reads_per_gpu": n, "reads_per_step_all_ranks": int(nloc.item()), "read_len": args.read_len, "min_len": args.min_len,
                "min_occ": args.min_occ, "mode": args.mode, "tags": not args.no_tags, "n_read_frac": args.n_read_frac,
                "sharding": ("reads sharded by rank, index replicated, no collective" if args.scaling == "weak" else
                             "ONE batch of %d reads cut into %d contiguous slices, one per rank (BASELINE configs[3] as worded), index replicated, no collective" % (args.reads, world)),
                "bwt_size": int(info.bwt_size), "bwt_runs": int(info.n_runs), "image_in_lds": bool(info.image_in_lds),
                "rank_image": "%s, %.1f MB%s" % (image, info.image_bytes / 1e6, "" if info.image_in_lds else
                                                 (", global memory (fits the 256 MB memory-side cache)" if info.image_bytes < 240e6 else ", resident in HBM")),
                "image_kind": int(info.image_kind), "image_pairs": int(info.image_pairs), "image_wide": int(info.image_wide), "pairs_stride": int(info.pairs_stride),
                "tag_image_MB": info.tag_image_bytes / 1e6,
                "index_build_host_s": round(build_s, 1), "prep_s": round(prep_s, 1),
            },
            "mems_per_s": tot[0] * K / dt,
            "extensions_per_s": tot[2] * K / dt,
            "mems_per_step": tot[0], "positions_per_step": tot[1], "extensions_per_step": tot[2],
            "kernel_ms_per_step": k_ms,
            "speculative_runs": dict(zip(("sized_from_the_previous_run", "repeated_with_exact_sizes"), batch.spec_stats())),
            "pairs_kernel": {"used": bool(batch.timing().pairs_reads), "extensions_through_the_dense2_image": int(batch.timing().pairs_other_steps)},
            "roofline": roofline_record(info, len(cat), n, counts, k_ms, batch.timing(), args.workload, args.min_len, not args.no_tags),
        }
        if args.pcie:
            line["pcie_inclusive_reads_per_s"] = pcie_rate(idx, cat, offs, args, local)
            line["pcie_inclusive_note"] = "long-lived batch: pgx_batch_upload (H2D of reads + offsets), pgx_batch_run, pgx_batch_result (D2H of MEMs / positions into host arrays)"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, ri, None if args.no_tags else tags, cat, offs, args.min_len)
        if world == 1 and args.overlap and not args.no_overlap and args.steps >= 2:
            dt2, c2 = measure_two_in_flight(idx, cat, offs, local, args.min_len, args.min_occ, flags, args.steps, torch.cuda.synchronize)
            line["two_batches_in_flight"] = {
                "value": n * args.steps / dt2, "unit": "reads/s", "ms_per_step": 1e3 * dt2 / args.steps, "steps": args.steps,
                "same_counts_as_the_bench_steps": all(tuple(c) == tuple(counts) for c in c2),
                "note": "not the bench value: K steps over two resident batches of the same reads, each on its own stream with its own host thread (what "
                        "the CLI's device workers do with consecutive batches): one batch's compaction and tag stage run under the other's find_mems kernel",
            }
        resident_result = None
        if world == 1 and not args.no_parity:
            # the measured configuration itself under the oracle: the result of the last timed step, downloaded, against the oracle on a prefix
            sample = min(n, args.parity_reads)
            resident_result = batch.result()
            line["parity_sample"] = parity_sample(args, ri, None if args.no_tags else tags, cat, offs, args.min_len, resident_result, sample, idx, local)
            parity_failed = parity_failed or not line["parity_sample"]["identical"]
        if world == 1 and not args.no_fresh and args.steps >= 3:
            # what a caller that brings a NEW batch every step gets (the bench value re-runs reads already resident in HBM)
            hbs = [HostBatch(P, cat, offs)] + [HostBatch(P, *make_workload.resample(k)) for k in (1, 2)]
            line["fresh_batch"] = fresh_record(P, idx, hbs, local, args, flags, torch.cuda.synchronize, resident_result, counts)
            for key in ("packed", "bytes"):
                if line["fresh_batch"][key]["identical_to_the_resident_batch"] is False:
                    parity_failed = True
            del hbs
        resident_result = None
        if world == 1 and args.batch_sweep:
            line["batch_size_sweep"] = batch_size_sweep(P, idx, cat, offs, local, args, flags, stream, torch.cuda.synchronize)
    batch.free()
    idx.close()
    del cat, offs
    if rank == 0:
        if args.workload == "chr22" and world == 1 and not args.no_secondary:
            line["secondary"] = secondary_x(args, P, wd, local, stream, torch)
            parity_failed = parity_failed or not line["secondary"].get("parity_sample", {}).get("identical", True)
        print(json.dumps(line), flush=True)
    barrier()
    if use_dist:
        dist.destroy_process_group()
    if own_tmp and rank == 0:
        import shutil

        shutil.rmtree(own_tmp, ignore_errors=True)
    if parity_failed:
        sys.stderr.write("[bench] PARITY MISMATCH between the device result and the oracle on the sample\n")
        sys.exit(3)


def pcie_rate(idx, cat, offs, args, local, min_len=None, reps=5):
    """host buffers in, host arrays out on a long-lived batch (device and pinned buffers reused): upload + run + download per call"""
    ml = args.min_len if min_len is None else min_len
    flags = 0 if args.no_tags else 1
    b = idx.batch(cat, offs, device=local)
    b.run(ml, args.min_occ, flags)
    b.result_counts()
    t1 = time.perf_counter()
    for _ in range(reps):
        b.upload(cat, offs)
        b.run(ml, args.min_occ, flags)
        b.result_counts()  # device -> pinned host arrays of the batch (what a C caller gets; no numpy copies on top)
    dt = time.perf_counter() - t1
    b.free()
    return (len(offs) - 1) * reps / dt


def secondary_x(args, P, wd, local, stream, torch):
    """BASELINE configs[1] (x index, 1 M reads, min_len 10: where the >= 10x CPU target is stated) measured in the same run"""
    n, min_len = DEFAULTS["x"][0], DEFAULTS["x"][1]
    ri, tags, cat, offs, desc, _ = make_workload(args, "x", 0, wd, lambda: None, n, None)
    idx = P.Index(ri, None if args.no_tags else tags, mode=P.MODE_COMPAT if args.mode == "compat" else P.MODE_STRICT)
    info = idx.info()
    flags = P.RUN_TIMING | (0 if args.no_tags else P.RUN_TAGS)
    batch, dt, k_ms, counts = measure(P, idx, cat, offs, local, min_len, args.min_occ, flags, args.steps, args.warmup, stream,
                                      torch.cuda.synchronize, lambda: None)
    out = {
        "workload": "BASELINE configs[1]: x.rl_bwt index (n=3012, sigma=5, no N), %d synthetic %d-bp reads, 1 MI355X" % (n, args.read_len),
        "value": n * args.steps / dt, "unit": "reads/s", "ms_per_step": dt / args.steps * 1e3, "min_len": min_len,
        "mems_per_step": counts[0], "positions_per_step": counts[1], "extensions_per_step": counts[2], "kernel_ms_per_step": k_ms,
        "image_in_lds": bool(info.image_in_lds),
        "roofline": roofline_record(info, len(cat), n, counts, k_ms, batch.timing(), "x", min_len, not args.no_tags),
        "pcie_inclusive_reads_per_s": pcie_rate(idx, cat, offs, args, local, min_len=min_len),
        "caveat": "no-N index: COMPAT searches die at every T (SURVEY 8a quirk 1), reads/s is inflated relative to a proper FMD index",
    }
    if not args.no_parity:
        out["parity_sample"] = parity_sample(args, ri, None if args.no_tags else tags, cat, offs, min_len, batch.result(), min(n, args.parity_reads), idx, local)
    batch.free()
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, ri, None if args.no_tags else tags, cat, offs, min_len, seconds=min(args.cpu_seconds, 8.0))
        out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["value"]
    idx.close()
    return out


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

    # the exchange: libpgx's own RCCL path (pgx_exchange_mems: device-resident records, count-sized broadcasts, device
    # interleave kernel) under the nccl backend; the torch / gloo implementation of pgx_shard only in CPU rehearsals
    comm, owner_of = None, [0] * K
    for r, cs in enumerate(owners):
        for c in cs:
            owner_of[c] = r
    if dev == "cuda":
        uid = [P.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        comm = P.Comm(uid[0], rank, world, device=local)

    def step():
        for c in owners[rank]:
            batches[c].run(args.min_len, args.min_occ, 0, stream)
        if comm is not None:
            n_r, n_m = comm.exchange({c: batches[c] for c in owners[rank]}, owner_of, download=False)
            return [n_m], None, None
        parts = {c: batches[c].result() for c in owners[rank]}
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
                       "exchange": ("pgx_exchange_mems: RCCL all-gather of u32 offsets + count-sized broadcasts of device-resident records + device interleave"
                                    if comm is not None else "torch.distributed all_gather (%s), host-staged records" % args.dist_backend)},
            "mems_per_step": int(mo[-1]), "mems_per_s": float(mo[-1]) * args.steps / dt,
        }), flush=True)
    for b in batches.values():
        b.free()
    barrier()
    if "WORLD_SIZE" in os.environ:
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


def cpu_baseline(args, ri, tags, cat, offs, min_len, seconds=None):
    """The oracle (kind "port": the reference cannot be built here) on a bounded sample of the same
    reads, all host cores via OpenMP over reads; timed regions = find_all_mems + tag queries."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O

    seconds = args.cpu_seconds if seconds is None else seconds
    r, t = oracle_objects(ri, tags)
    mode = O.MODE_COMPAT if args.mode == "compat" else O.MODE_STRICT
    cores = host_cores(O.lib().orc_max_threads())
    L = args.read_len
    n_reads = len(offs) - 1
    # one thread first (the reference's own execution model): calibrates the per-core rate
    one = min(n_reads, 5000)
    res1 = O.find_mems_batch(r, t, cat[: one * L], offs[: one + 1], min_len, args.min_occ, mode=mode, threads=1)
    sec1 = max(res1["seconds_mems"] + res1["seconds_tags"], 1e-6)
    # all cores on a sample worth ~cpu_seconds core-seconds (bounded by the batch), 3 runs, median
    sample = int(min(n_reads, max(one, (one / sec1) * seconds)))
    secs = []
    for _ in range(3):
        res = O.find_mems_batch(r, t, cat[: sample * L], offs[: sample + 1], min_len, args.min_occ, mode=mode, threads=cores)
        secs.append(res["seconds_mems"] + res["seconds_tags"])
    med = sorted(secs)[1]
    return {"value": sample / med, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d reads of the same batch (~%.0f core-seconds), OpenMP schedule(dynamic,256) over reads, "
                      "find_all_mems + tag-query compute only, 3 runs, median (%.3f s wall; all three: %s)"
                      % (sample, sample / (one / sec1), med, ", ".join("%.3f" % v for v in secs)),
            "single_thread_reads_per_s": one / sec1, "single_thread_sample": one}


if __name__ == "__main__":
    main()

"""Read sharding across the GPUs of one node: reads are independent
(find_all_mems carries no cross-read state, algorithm.hpp:739-757), the index is replicated, and
rank r owns one contiguous slice of the batch.  There is no data-path collective: per-rank CSR
results are concatenated in rank order, which is bit-identical to the unsharded result."""
import numpy as np


def shard_bounds(n_reads, world):
    """contiguous slices [lo, hi) per rank, sizes differing by at most one"""
    base, extra = divmod(n_reads, world)
    lo = [r * base + min(r, extra) for r in range(world)]
    return [(lo[r], lo[r] + base + (1 if r < extra else 0)) for r in range(world)]


def shard_reads(reads_cat, offsets, rank, world):
    """slice of a CSR read batch for one rank (offsets stay absolute; the C ABI rebases them)"""
    lo, hi = shard_bounds(len(offsets) - 1, world)[rank]
    return reads_cat, offsets[lo:hi + 1]


def merge_results(parts):
    """concatenate per-rank results (dicts as returned by pgx_ffi.Batch.result) in rank order"""
    out = {}
    mo, po = [np.zeros(1, np.uint64)], [np.zeros(1, np.uint64)]
    mbase = pbase = np.uint64(0)
    for p in parts:
        mo.append(p["mem_offsets"][1:] + mbase)
        mbase = mbase + p["mem_offsets"][-1]
        if "pos_offsets" in p:
            po.append(p["pos_offsets"][1:] + pbase)
            pbase = pbase + p["pos_offsets"][-1]
    out["mem_offsets"] = np.concatenate(mo)
    out["mems"] = np.concatenate([p["mems"] for p in parts])
    out["n_extensions"] = sum(int(p["n_extensions"]) for p in parts)
    if all("pos_offsets" in p for p in parts):
        out["pos_offsets"] = np.concatenate(po)
        out["tag_run_counts"] = np.concatenate([p["tag_run_counts"] for p in parts])
        out["positions"] = np.concatenate([p["positions"] for p in parts])
        out["n_tag_overflow"] = sum(int(p.get("n_tag_overflow", 0)) for p in parts)
    return out


# ------------------------------------------------------------------------------------------------
# Chromosome-sharded mode (BASELINE configs[4]): every rank holds the indexes of a subset
# of chromosomes, every read is searched in every shard, and the per-read result is the concatenation,
# in chromosome order, of the per-shard MEM lists (each bit-exact with the reference run on that
# shard's index; this is NOT the MEM set of a merged whole-genome index -- different maximality,
# different BWT coordinates, min_occ applies per shard).  This is the one place the path has a real
# exchange step: per-read counts (u32) and 32-byte MEM records travel between ranks with
# all_gather over torch.distributed ("nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).
def lpt_assign(lengths, world):
    """longest-processing-time bin packing of chromosomes onto ranks; returns chrom ids per rank (sorted)"""
    order = sorted(range(len(lengths)), key=lambda c: (-lengths[c], c))
    load, bins = [0] * world, [[] for _ in range(world)]
    for c in order:
        r = min(range(world), key=lambda k: (load[k], k))
        bins[r].append(c)
        load[r] += lengths[c]
    return [sorted(b) for b in bins]


def exchange_mems(local, n_reads, n_chroms, dist=None, device="cpu"):
    """local: {chrom id: result dict with 'mem_offsets' (n_reads+1) and 'mems' (MEM_DTYPE)} of THIS rank.
    Returns the merged CSR (mem_offsets, mems, shard_of_mem) -- identical on every rank.

    Collectives: all_gather of the per-read counts (padded to the largest local chromosome count) and
    all_gather of the MEM records (padded to the largest per-rank total); payload ~ MEMs x 32 B."""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    mine = sorted(local)
    # small metadata: which chromosomes each rank holds
    owners = [mine]
    if dist is not None:
        owners = [None] * world
        dist.all_gather_object(owners, mine)
    max_local = max(1, max(len(o) for o in owners))
    counts = torch.zeros((max_local, n_reads), dtype=torch.int64, device=device)
    recs = []
    for k, c in enumerate(mine):
        if local[c].get("device"):  # Batch.device_result(): the arrays are already in HBM, no host staging
            mo = torch.as_tensor(local[c]["mem_offsets"], device=device)
            counts[k] = mo[1:] - mo[:-1]
            recs.append(torch.as_tensor(local[c]["mems"], device=device) if local[c]["n_mems"] else torch.zeros((0, 4), dtype=torch.int64, device=device))
        else:
            mo = local[c]["mem_offsets"].astype(np.int64)
            counts[k] = torch.from_numpy(mo[1:] - mo[:-1]).to(device)
            recs.append(torch.from_numpy(np.ascontiguousarray(local[c]["mems"]).view(np.int64).reshape(-1, 4).copy()).to(device))
    rec = torch.cat(recs) if recs else torch.zeros((0, 4), dtype=torch.int64, device=device)
    m_local = torch.tensor([rec.shape[0]], dtype=torch.int64, device=device)
    if dist is not None:
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts)
        all_m = [torch.zeros_like(m_local) for _ in range(world)]
        dist.all_gather(all_m, m_local)
        m_max = max(1, int(max(int(t.item()) for t in all_m)))
        padded = torch.zeros((m_max, 4), dtype=torch.int64, device=device)
        padded[: rec.shape[0]] = rec
        all_rec = [torch.zeros_like(padded) for _ in range(world)]
        dist.all_gather(all_rec, padded)
    else:
        all_counts, all_rec = [counts], [rec]
    # merged offsets: total MEMs per read over all chromosomes
    per_chrom = {}
    for r, o in enumerate(owners):
        for k, c in enumerate(o):
            per_chrom[c] = (r, k)
    assert sorted(per_chrom) == list(range(n_chroms)), "every chromosome must be owned by exactly one rank"
    total = torch.zeros(n_reads, dtype=torch.int64, device=device)
    for c in range(n_chroms):
        r, k = per_chrom[c]
        total += all_counts[r][k]
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
    offs[1:] = torch.cumsum(total, 0)
    out = torch.zeros((int(offs[-1].item()), 4), dtype=torch.int64, device=device)
    shard = torch.zeros(int(offs[-1].item()), dtype=torch.int64, device=device)
    before = torch.zeros(n_reads, dtype=torch.int64, device=device)  # MEMs of earlier chromosomes, per read
    src_base = [0] * world  # running offset into each rank's concatenated records (chrom order = owner order)
    ar = torch.arange(n_reads, device=device)
    for c in range(n_chroms):
        r, k = per_chrom[c]
        cnt = all_counts[r][k]
        m = int(cnt.sum().item())
        if m:
            read_of = torch.repeat_interleave(ar, cnt)
            excl = torch.cumsum(cnt, 0) - cnt
            within = torch.arange(m, device=device) - excl[read_of]
            dst = offs[:-1][read_of] + before[read_of] + within
            out[dst] = all_rec[r][src_base[r]: src_base[r] + m]
            shard[dst] = c
        src_base[r] += m
        before += cnt
    from pgx_ffi import MEM_DTYPE

    mems = out.cpu().numpy().astype(np.int64).reshape(-1).view(MEM_DTYPE) if out.shape[0] else np.zeros(0, MEM_DTYPE)
    return offs.cpu().numpy().astype(np.uint64), mems, shard.cpu().numpy()

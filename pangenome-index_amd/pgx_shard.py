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

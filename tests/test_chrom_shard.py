"""Chromosome-sharded mode (SURVEY 8e row 2 / BASELINE configs[4]): per-read concatenation, in shard order,
of per-shard MEM lists, exchanged with all_gather.  CPU tier: gloo world-size 2 with the oracle as the
per-shard engine; GPU tier: the HIP path as the engine (single process, no collective)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_ffi as O
import pgx_shard as S
import pgx_workload as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = O.GOLDEN
CHROMS = [("bidirectional_test/contigs_xy.rl_bwt", "bidirectional_test/contigs_xy"), ("x.rl_bwt", "x.newline_separated"),
          ("two_contig_graph/contigs_XY.rl_bwt", "two_contig_graph/contigs_XY.txt")]


def _expected(indexes, cat, offs, min_len, min_occ):
    """definition of the mode: for every read, shard 0's MEMs, then shard 1's, ..."""
    parts = [O.find_mems_batch(ri, None, cat, offs, min_len, min_occ) for ri in indexes]
    mems, mo, shard = [], [0], []
    for i in range(len(offs) - 1):
        for c, p in enumerate(parts):
            seg = p["mems"][p["mem_offsets"][i]:p["mem_offsets"][i + 1]]
            mems.append(seg)
            shard += [c] * len(seg)
        mo.append(len(shard))
    return np.array(mo, dtype=np.uint64), np.concatenate(mems), np.array(shard), parts


def _setup(workdir):
    idx = []
    for k, (rl, _) in enumerate(CHROMS):
        ri, _t = W.build_index_from_rlbwt(os.path.join(G, rl), workdir, "chrom%d" % k, with_tags=False)
        idx.append(ri)
    seqs = []
    for _, t in CHROMS:
        seqs += W.load_sequences(os.path.join(G, t))
    cat, offs = W.sample_reads(seqs, 1500, 100, seed=8)
    return idx, cat, offs


def test_lpt_assign():
    a = S.lpt_assign([10, 9, 8, 7, 6, 5, 4, 3], 3)
    assert sorted(c for b in a for c in b) == list(range(8))
    loads = [sum([10, 9, 8, 7, 6, 5, 4, 3][c] for c in b) for b in a]
    assert max(loads) <= (4 * sum(loads)) // (3 * 3) + 1  # LPT bound: 4/3 of the ideal load
    assert S.lpt_assign([5, 5], 4) == [[0], [1], [], []]


def _meta_rows(world, owner, n_reads, mems_per_shard, status=None):
    """what the metadata all-gather of pgx_exchange_mems delivers: one row per rank"""
    import pgx_ffi as P

    per_rank = [[c for c, o in enumerate(owner) if o == r] for r in range(world)]
    ml = max(1, max(len(x) for x in per_rank))
    rows = np.zeros((world, P.XCH_META_HEAD + ml), dtype=np.uint64)
    for r in range(world):
        rows[r, 0] = (status or {}).get(r, 0)
        rows[r, 1] = n_reads if per_rank[r] else 0  # a rank without shards has searched nothing
        rows[r, 2] = len(owner)
        rows[r, 3] = P.exchange_owner_digest(owner)
        rows[r, 4] = ml
        for k, c in enumerate(per_rank[r]):
            rows[r, P.XCH_META_HEAD + k] = mems_per_shard[c]
    return rows, ml


def test_exchange_plan_uneven_and_zero_shard_owners(built):
    """the host-side plan of pgx_exchange_mems (slots, record bases) for ranks with different shard counts and ranks without shards:
    the RCCL path with more than one rank cannot run on the one-GPU box, its arithmetic can"""
    import pgx_ffi as P

    owner = [2, 0, 2, 2, 0, 3]  # rank 1 owns nothing, rank 2 three shards, rank 0 two, rank 3 one
    mems = [7, 0, 11, 5, 3, 2]
    rows, ml = _meta_rows(4, owner, 1000, mems)
    plan = P.exchange_plan(4, owner, rows)
    assert plan["max_local"] == ml == 3 and plan["n_reads"] == 1000
    # rank r's k-th shard (ascending shard id) sits in row r * max_local + k
    assert list(plan["slot"]) == [2 * 3 + 0, 0 * 3 + 0, 2 * 3 + 1, 2 * 3 + 2, 0 * 3 + 1, 3 * 3 + 0]
    # records: rank 0's (shards 1, 4), rank 1's (none), rank 2's (0, 2, 3), rank 3's (5)
    assert list(plan["rec_base"]) == [0, 0 + 3, 3, 3 + 7 + 11 + 5, 26 + 2]
    assert list(plan["src_base"]) == [3, 0, 3 + 7, 3 + 7 + 11, 0, 26]
    # every rank computes the same plan from the same rows; a one-rank world is the degenerate case the GPU test runs
    rows1, _ = _meta_rows(1, [0, 0, 0], 5, [1, 2, 3])
    p1 = P.exchange_plan(1, [0, 0, 0], rows1)
    assert list(p1["slot"]) == [0, 1, 2] and list(p1["src_base"]) == [0, 1, 3] and list(p1["rec_base"]) == [0, 6]


def test_exchange_plan_failures_are_collective(built):
    import pgx_ffi as P

    owner = [0, 1, 1]
    rows, _ = _meta_rows(2, owner, 10, [1, 1, 1], status={1: P.ERR_ARG})
    with pytest.raises(P.PgxError) as e:  # one rank's local validation failed: every rank sees its status word and fails
        P.exchange_plan(2, owner, rows)
    assert "rank 1 reported status" in str(e.value)
    rows, _ = _meta_rows(2, owner, 10, [1, 1, 1])
    rows[1, 1] = 11
    with pytest.raises(P.PgxError) as e:
        P.exchange_plan(2, owner, rows)
    assert "disagree on the number of reads" in str(e.value)
    rows, _ = _meta_rows(2, owner, 10, [1, 1, 1])
    rows[0, 3] ^= 1  # a rank that was handed another owner table
    with pytest.raises(P.PgxError) as e:
        P.exchange_plan(2, owner, rows)
    assert "different owner_of_shard" in str(e.value)
    # the gathered offsets are world x max_local x (n_reads + 1) u32: the caller must chunk its reads
    big = P.XCH_MAX_OFFSET_BYTES // (2 * 2 * 4) + 1
    rows, _ = _meta_rows(2, owner, big, [1, 1, 1])
    with pytest.raises(P.PgxError) as e:
        P.exchange_plan(2, owner, rows)
    assert "in chunks" in str(e.value)
    with pytest.raises(P.PgxError):
        P.exchange_plan(2, [0, 5], np.zeros((2, 6), dtype=np.uint64))  # owner rank out of range


def test_exchange_single_process(workdir):
    paths, cat, offs = _setup(workdir)
    indexes = [O.RIndex(p) for p in paths]
    mo, mems, shard, parts = _expected(indexes, cat, offs, 5, 1)
    got = S.exchange_mems({c: parts[c] for c in range(3)}, len(offs) - 1, 3)
    assert np.array_equal(got[0], mo) and got[1].tobytes() == mems.tobytes() and np.array_equal(got[2], shard)
    assert len(mems) > 1000 and len(set(shard)) == 3


WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "pangenome-index_amd"))
import numpy as np, torch, torch.distributed as dist
import oracle_ffi as O, pgx_shard as S
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from test_chrom_shard import _setup, _expected
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
paths, cat, offs = _setup(%(wd)r + "/r%%d" %% rank)
indexes = [O.RIndex(p) for p in paths]
mine = S.lpt_assign([8022, 3012, 4011], world)[rank]           # chromosome lengths -> ranks
local = {c: O.find_mems_batch(indexes[c], None, cat, offs, 5, 1) for c in mine}
got = S.exchange_mems(local, len(offs) - 1, 3, dist=dist, device="cpu")
mo, mems, shard, _ = _expected(indexes, cat, offs, 5, 1)
ok = np.array_equal(got[0], mo) and got[1].tobytes() == mems.tobytes() and np.array_equal(got[2], shard)
print("RANK%%d %%s %%d" %% (rank, "CHROM_OK" if ok else "CHROM_MISMATCH", len(mems)), flush=True)
dist.destroy_process_group()
'''


def test_exchange_world_size_2_gloo(built, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "wd": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29519", str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert out.stdout.count("CHROM_OK") == 2, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.gpu
def test_exchange_through_the_c_abi_rccl(workdir):
    """pgx_exchange_mems: RCCL all-gather of the u32 offsets + one broadcast per rank of the records + the device interleave
    kernel, here with a communicator of one rank that owns all three shards (the GPU box has one device; the same entry point
    runs the N-rank case of an 8-GPU node), against the definition of the mode computed with the oracle.  In a process of its
    own: RCCL comes with its own runtime state, which should not meet the torch imported by other tests of this module."""
    paths, cat, offs = _setup(workdir)
    indexes = [O.RIndex(p) for p in paths]
    exp = os.path.join(workdir, "xch_expected.npz")
    z = {"cat": cat, "offs": offs, "paths": np.array(paths)}
    for min_len in (5, 12):
        mo, mems, shard, _ = _expected(indexes, cat, offs, min_len, 1)
        z.update({"mo%d" % min_len: mo, "mems%d" % min_len: mems.view(np.int64), "shard%d" % min_len: shard})
    np.savez(exp, **z)
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import pgx_ffi as P
z = np.load(%r)
paths, cat, offs = [str(p) for p in z["paths"]], z["cat"], z["offs"]
for min_len in (5, 12):
    mo, mems, shard = z["mo%%d" %% min_len], z["mems%%d" %% min_len], z["shard%%d" %% min_len]
    idx = [P.Index(p) for p in paths]
    batches = {c: idx[c].batch(cat, offs) for c in range(3)}
    for b in batches.values():
        b.run(min_len, 1, 0)
    comm = P.Comm(P.comm_unique_id(), 0, 1)
    got = comm.exchange(batches, [0, 0, 0])
    assert np.array_equal(got[0], mo) and np.array_equal(got[1].view(np.int64), mems) and np.array_equal(got[2], shard.astype(np.uint32))
    # a second exchange on the same communicator (buffers are reused) with the shards given in another order
    got = comm.exchange({2: batches[2], 0: batches[0], 1: batches[1]}, [0, 0, 0])
    assert np.array_equal(got[0], mo) and np.array_equal(got[1].view(np.int64), mems)
    try:
        comm.exchange({0: batches[0]}, [0, 0, 0])  # this rank owns three shards
        raise SystemExit("no error for a missing shard")
    except P.PgxError:
        pass
    comm.free()
    for b in batches.values():
        b.free()
    for i in idx:
        i.close()
    print("rccl exchange ok", min_len, len(mems))
""" % (os.path.join(ROOT, "pangenome-index_amd"), exp)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.count("rccl exchange ok") == 2, r.stdout + r.stderr


@pytest.mark.gpu
def test_exchange_with_hip_engine(workdir):
    import pgx_ffi as P

    paths, cat, offs = _setup(workdir)
    indexes = [O.RIndex(p) for p in paths]
    mo, mems, shard, _ = _expected(indexes, cat, offs, 5, 1)
    local = {c: P.Index(paths[c]).find_mems(cat, offs, 5, 1) for c in range(3)}
    got = S.exchange_mems(local, len(offs) - 1, 3)
    assert np.array_equal(got[0], mo) and got[1].tobytes() == mems.tobytes() and np.array_equal(got[2], shard)
    # device-resident variant: the batches' device buffers go into the exchange without host staging (what the nccl path
    # does).  In its own process with torch imported first, as bench.py does: torch's bundled HIP runtime does not
    # initialise once the system runtime behind libpgx.so owns the device.
    exp = os.path.join(workdir, "chrom_expected.npz")
    np.savez(exp, mo=mo, mems=mems.view(np.int64), shard=shard, cat=cat, offs=offs, paths=np.array(paths))
    code = """
import sys, numpy as np, torch
torch.cuda.init()
sys.path.insert(0, %r); sys.path.insert(0, %r)
import pgx_ffi as P, pgx_shard as S
z = np.load(%r)
paths, cat, offs = [str(p) for p in z["paths"]], z["cat"], z["offs"]
idx = [P.Index(p) for p in paths]
batches = [i.batch(cat, offs) for i in idx]
for b in batches:
    b.run(5, 1, 0)
dev_local = {c: batches[c].device_result() for c in range(3)}
assert all(d["device"] for d in dev_local.values())
got = S.exchange_mems(dev_local, len(offs) - 1, 3, device="cuda")
assert np.array_equal(got[0], z["mo"]) and np.array_equal(got[1].view(np.int64), z["mems"]) and np.array_equal(got[2], z["shard"])
print("device exchange ok", sum(d["n_mems"] for d in dev_local.values()))
""" % (os.path.dirname(os.path.abspath(P.__file__)), os.path.dirname(os.path.abspath(__file__)), exp)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "device exchange ok" in r.stdout, r.stdout + r.stderr

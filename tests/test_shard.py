"""CPU tier, world_size 2 over gloo: the N>1 path of bench.py / pgx_shard (reads sharded by rank,
index replicated, no data-path collective; per-rank CSR results concatenated in rank order).  The
per-shard engine here is the oracle (there is no GPU in this tier); on the GPU box the same merge
is exercised with the HIP path by tests/test_gpu_parity.py::test_sharded_equals_unsharded."""
import os
import subprocess
import sys

import numpy as np

import oracle_ffi as O
import pgx_shard as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "pangenome-index_amd"))
import numpy as np, torch, torch.distributed as dist
import oracle_ffi as O, pgx_shard as S, pgx_workload as W
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
G = O.GOLDEN
ri = O.RIndex(os.path.join(G, "bidirectional_test", "xy.ri"))
tags = O.Tags(os.path.join(G, "bidirectional_test", "xy_bidirectional_compressed.tags"), O.TAGS_BYTECODE)
seqs = W.load_sequences(os.path.join(G, "bidirectional_test", "contigs_xy"))
cat, offs = W.sample_reads(seqs, 3001, 100, seed=5)          # same batch on every rank
cat_r, offs_r = S.shard_reads(cat, offs, rank, world)         # this rank's contiguous slice
part = O.find_mems_batch(ri, tags, cat_r, offs_r - offs_r[0] + offs_r[0], 5, 1)
dist.barrier()
t = torch.tensor([float(len(part["mems"]))], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.SUM)                       # the only collective bench.py uses: counters
parts = [None] * world
dist.all_gather_object(parts, part)
if rank == 0:
    merged = S.merge_results(parts)
    whole = O.find_mems_batch(ri, tags, cat, offs, 5, 1)
    ok = (np.array_equal(merged["mem_offsets"], whole["mem_offsets"]) and merged["mems"].tobytes() == whole["mems"].tobytes()
          and np.array_equal(merged["pos_offsets"], whole["pos_offsets"]) and np.array_equal(merged["positions"], whole["positions"])
          and np.array_equal(merged["tag_run_counts"], whole["tag_run_counts"]) and merged["n_extensions"] == whole["n_extensions"]
          and int(t.item()) == len(whole["mems"]) and len(whole["mems"]) > 0)
    print("SHARD_OK" if ok else "SHARD_MISMATCH", len(whole["mems"]))
dist.destroy_process_group()
'''


def test_shard_bounds():
    for n in (0, 1, 2, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            b = S.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gloo(built, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29517", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert "SHARD_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]

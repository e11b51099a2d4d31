"""bench.py --gpus N starts N ranks by itself (no external launcher) and every rank joins the process group: checked on the
CPU with the gloo backend and a stub workload (the real workloads need a GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=300, env=e)
    return r


def test_gpus_flag_fans_out_two_ranks():
    r = _run(["--gpus", "2", "--stub-workload", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]  # (gloo itself may print a connection notice)
    assert len(lines) == 1  # rank 0 prints ONE JSON line, the other ranks nothing
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1


def test_single_rank_needs_no_process_group():
    r = _run(["--stub-workload"])
    assert r.returncode == 0, r.stderr
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["n_ranks_seen"] == 1


def test_external_launcher_environment_is_respected():
    """under torch.distributed.run (RANK / WORLD_SIZE already set) bench.py must not start ranks of its own"""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        e = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-workload", "--dist-backend", "gloo"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
    assert json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])["n_ranks_seen"] == 2
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]


def test_failing_rank_fails_the_launch():
    r = _run(["--gpus", "2", "--workload", "x", "--dist-backend", "gloo", "--one-device", "--steps", "1", "--warmup", "0"])
    # no GPU in the CPU tier: every rank exits non-zero ("bench.py needs a GPU"), and so must the launcher
    if r.returncode == 0:  # a GPU box: the run really works there
        assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 2
    else:
        assert "ranks failed" in r.stderr


def test_strong_scaling_cuts_one_batch_into_contiguous_slices():
    """bench.py --scaling strong (BASELINE configs[3] as worded: ONE batch read-sharded over the ranks; reference unit: the per-read loop
    src/find_mems.cpp:94-139 over one reads file): the ranks' slices tile the batch; weak scaling gives every rank a batch of its own"""
    r = _run(["--gpus", "2", "--stub-workload", "--dist-backend", "gloo", "--scaling", "strong", "--reads", "1001", "--read-len", "7"])
    assert r.returncode == 0, r.stderr
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert rec["scaling"] == "strong" and rec["reads_per_rank"] == [500, 501] and rec["first_read_per_rank"] == [0, 500]
    assert rec["read_bytes_per_rank"] == [3500, 3507]
    r = _run(["--gpus", "2", "--stub-workload", "--dist-backend", "gloo", "--reads", "1001", "--read-len", "7"])
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert rec["scaling"] == "weak" and rec["reads_per_rank"] == [1001, 1001]
    sys.path.insert(0, ROOT)
    import numpy as np

    import bench

    offs = np.array([0, 3, 3, 10, 12, 20], dtype=np.uint64)
    cat = np.arange(20, dtype=np.uint8)
    parts = [bench.strong_slice(cat, offs, k, 3) for k in range(3)]
    assert [len(o) - 1 for _, o in parts] == [1, 2, 2] and all(int(o[0]) == 0 for _, o in parts)
    assert np.array_equal(np.concatenate([c for c, _ in parts]), cat)
    assert [int(o[-1]) for _, o in parts] == [3, 7, 10]

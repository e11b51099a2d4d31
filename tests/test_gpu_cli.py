"""GPU tier: the find_mems CLI is a drop-in for the reference's process contract
(`find_mems <ri> <tags> <reads> <min_len> <min_occ>`, stdout grammar, exit codes), and the C++
compatibility headers (FastLocate / TagArray / find_all_mems) give the same output read by read."""
import os
import subprocess

import numpy as np
import pytest

import oracle_ffi as O
import pgx_workload as W
from cli_format import format_find_mems, strip_timing

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "pangenome-index_amd", "find_mems")
DEMO = os.path.join(ROOT, "pangenome-index_amd", "compat_demo")
BT = os.path.join(O.GOLDEN, "bidirectional_test")


def _run(exe, *args, env=None):
    return subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))


@pytest.mark.parametrize("reads_file,ml,mo", [("reads.txt", 5, 1), ("reads.txt", 3, 1), ("test_reads.txt", 3, 1)])
def test_cli_matches_committed_golden(built, reads_file, ml, mo):
    r = _run(CLI, os.path.join(BT, "xy.ri"), os.path.join(BT, "xy_bidirectional_compressed.tags"), os.path.join(BT, reads_file), ml, mo)
    assert r.returncode == 0, r.stderr
    exp = open(os.path.join(O.GOLDEN, "expected_find_mems_xy_%s_%d_%d.txt" % (reads_file.split(".")[0], ml, mo))).read()
    assert strip_timing(r.stdout) == exp
    n_reads = len([l for l in open(os.path.join(BT, reads_file)).read().split("\n") if l])
    assert r.stderr.count("[find_all_mems] total mems=") == n_reads
    assert "Reading the rindex file (encoded)" in r.stderr and "Reading the tag array index" in r.stderr


def test_cli_many_reads_small_batches(built, x_index, workdir):
    """multi-batch path (--batch) keeps file order and the 1-based Seq counter; empty lines are skipped"""
    ri, tags = x_index
    seqs = W.load_sequences(os.path.join(O.GOLDEN, "x.newline_separated"))
    cat, offs = W.sample_reads(seqs, 5000, 150, seed=77)
    path = os.path.join(workdir, "reads5000.txt")
    with open(path, "w") as f:
        for i in range(5000):
            f.write(bytes(cat[offs[i]:offs[i + 1]]).decode() + "\n")
            if i % 7 == 0:
                f.write("\n")
    ref = O.find_mems_batch(O.RIndex(ri), O.Tags(tags, O.TAGS_COMPACT), cat, offs, 10, 1, threads=4)
    exp = format_find_mems(ref)
    # --devices 0,0: two device slots (here the same GPU twice), each with its own worker threads, batches and streams: the
    # path `--gpus N` takes on an N-GPU node; the text must come out in file order whichever worker finishes first
    for extra in ([], ["--batch", "777", "--quiet"], ["--tags-format", "compact", "--device", "0"],
                  ["--devices", "0,0", "--streams", "2", "--batch", "300"], ["--gpus", "1", "--streams", "1", "--batch", "5000"],
                  ["--devices", "0,0,0", "--streams", "1", "--batch", "64", "--quiet"]):
        r = _run(CLI, ri, tags, path, 10, 1, *extra)
        assert r.returncode == 0, r.stderr
        assert strip_timing(r.stdout) == exp
    # PGX_CLI_PACKED=1: the parse threads pack the reads (pgx_pack_reads) and the batches travel packed (pgx_batch_upload_packed): the same text
    for extra in (["--batch", "777", "--quiet"], ["--devices", "0,0", "--streams", "2", "--batch", "300"]):
        r = _run(CLI, ri, tags, path, 10, 1, *extra, env={"PGX_CLI_PACKED": "1"})
        assert r.returncode == 0, r.stderr
        assert strip_timing(r.stdout) == exp


def test_compat_headers_same_output(built):
    args = (os.path.join(BT, "xy.ri"), os.path.join(BT, "xy_bidirectional_compressed.tags"), os.path.join(BT, "reads.txt"), 5, 1)
    a, b = _run(CLI, *args), _run(DEMO, *args)
    assert a.returncode == 0 and b.returncode == 0, b.stderr
    assert strip_timing(a.stdout) == b.stdout
    # public members of FastLocate through the compat header: SURVEY 8c known answer A -> (8, 3420, 2309)
    assert "sigma=5 sym_map[A]=1 bwd(A)=8,3420,2309 comp(A)=T strings=8" in b.stderr
    # locate members: against the oracle's chain on the same file
    r = O.RIndex(os.path.join(BT, "xy.ri"))
    sa = r.decompress_sa()
    ml = r.max_length
    exp = "locate: first=%d next=%d seq(first)=%d+%d DA[0..3]=%d,%d,%d,%d |DA|=%d locate(endmarkers)=8:0..7" % (
        sa[0], sa[1], sa[0] // ml, sa[0] % ml, sa[0] // ml, sa[1] // ml, sa[2] // ml, sa[3] // ml, len(sa))
    assert exp in b.stderr, b.stderr


def test_cli_errors(built, workdir):
    r = _run(CLI, "/nonexistent.ri", "x", "y", 5, 1)
    assert r.returncode == 1 and "Cannot open r-index: /nonexistent.ri" in r.stderr  # find_mems.cpp:30
    r = _run(CLI, os.path.join(BT, "xy.ri"), os.path.join(BT, "xy_bidirectional_compressed.tags"), "/nonexistent.txt", 5, 1)
    assert r.returncode == 1 and "Cannot open reads file: /nonexistent.txt" in r.stderr  # find_mems.cpp:91
    bad = os.path.join(workdir, "foreign.ri")
    open(bad, "wb").write(b"\x00\x0a\x03\x00" + b"\x00" * 100)
    r = _run(CLI, bad, os.path.join(BT, "xy_bidirectional_compressed.tags"), os.path.join(BT, "reads.txt"), 5, 1)
    assert r.returncode == 1 and "FastLocate: Invalid tag" in r.stderr  # src/r-index.cpp:412-414
    assert _run(CLI).returncode == 1

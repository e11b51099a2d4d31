"""End-to-end wall time of the find_mems CLI (text in, text out) on the GPU box: python3 scripts/cli_e2e.py [x|synth] [n_reads] [check]
(check: the text written with one worker, with three workers and with two device slots is the same file, byte for byte)"""
import os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd"))
import numpy as np
import pgx_workload as W

wl = sys.argv[1] if len(sys.argv) > 1 else "x"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
wd = "/tmp/pgx_cli_e2e"
os.makedirs(wd, exist_ok=True)
if wl == "x":
    g = os.path.join(ROOT, "tests", "golden")
    ri, tags = W.build_index_from_rlbwt(os.path.join(g, "x.rl_bwt"), wd, "x")[:2]
    seqs = W.load_sequences(os.path.join(g, "x.newline_separated"))
    min_len = 10
else:
    text = os.path.join(wd, "synth.txt")
    W.synth_pangenome_text(text)
    ri, tags = W.build_index_from_text(text, wd, "synth")[:2]
    seqs = W.load_sequences(text)
    min_len = 20
cat, offs = W.sample_reads(seqs, n, 150, seed=42)
path = os.path.join(wd, "reads.txt")
lines = np.full((n, 151), 10, dtype=np.uint8)  # one read per line
lines[:, :150] = cat.reshape(n, 150)
lines.tofile(path)
print("reads file written", flush=True)
exe = os.path.join(ROOT, "pangenome-index_amd", "find_mems")
# the pipeline: 1 worker = upload, run, download and formatting of one batch after the other (the sum of the stages);
# 3 workers on one device = the stages of consecutive batches overlap; two device slots = what --gpus 2 does on two GPUs
for extra in (["--streams", "1", "--batch", "1048576"], ["--streams", "1"], ["--streams", "3"], ["--devices", "0,0", "--streams", "2"]):
    for dest in ("/dev/null",):
        t0 = time.time()
        with open(dest, "wb") as out:
            r = subprocess.run([exe, ri, tags, path, str(min_len), "1", "--quiet"] + extra, stdout=out, stderr=subprocess.PIPE, env=dict(os.environ, PGX_CLI_STATS="1"))
        dt = time.time() - t0
        print("%s n=%d %s -> %s: %.2f s wall (%.2f M reads/s end to end), rc=%d" % (wl, n, " ".join(extra), dest, dt, n / dt / 1e6, r.returncode))
        err = r.stderr.decode().strip().split("\n")
        print("   ", [l for l in err if "took" in l or "[find_mems]" in l], flush=True)

if len(sys.argv) > 3 and sys.argv[3] == "check":
    import hashlib
    sums = []
    for i, extra in enumerate((["--streams", "1", "--batch", "1048576"], ["--streams", "3"], ["--devices", "0,0", "--streams", "2", "--batch", "65536"])):
        dest = os.path.join(wd, "out%d.txt" % i)
        with open(dest, "wb") as out:
            r = subprocess.run([exe, ri, tags, path, str(min_len), "1", "--quiet"] + extra, stdout=out, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-400:]
        h = hashlib.md5()
        with open(dest, "rb") as f:
            # the two "Total time" lines at the end differ from run to run
            body = f.read()
        cut = body.rfind(b"Total time")
        cut = body.rfind(b"Total time", 0, cut)
        h.update(body[:cut])
        sums.append((h.hexdigest(), len(body)))
        print("%s %s: %d bytes, md5 of the body %s" % (wl, " ".join(extra), len(body), sums[-1][0]), flush=True)
    assert len({s for s, _ in sums}) == 1, sums
    print("check: identical output under every pipeline shape")

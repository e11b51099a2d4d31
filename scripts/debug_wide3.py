import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pgx_workload as W, pgx_ffi as P, oracle_ffi as O
wd = tempfile.mkdtemp()
text = os.path.join(wd, "w.txt")
W.synth_pangenome_text(text, base_len=30_000, n_hap=3, seed=77, snp=0.01, indel=0.001, n_runs=3, n_run_len=(20, 400))
ri, tags, _ = W.build_index_from_text(text, wd, "w")
o = O.RIndex(ri)
seqs = W.load_sequences(text)
cat, offs = W.sample_reads(seqs, 200_000, 150, seed=5)
ref = O.find_mems_batch(o, None, cat, offs, 20, 1, threads=8)
for shift in ("22", "2"):
    os.environ["PGX_SB_SHIFT"] = shift
    for force in (P.MODE_IMAGE_DENSE2, P.MODE_IMAGE_PAIRS):
        idx = P.Index(ri, None, mode=force | P.MODE_IMAGE_WIDE)
        for trial in range(3):
            res = idx.find_mems(cat, offs, 20, 1)
            ok = np.array_equal(res["mem_offsets"], ref["mem_offsets"]) and res["mems"].tobytes() == ref["mems"].tobytes() and res["n_extensions"] == ref["n_extensions"]
            nbad = -1
            if not ok and len(res["mems"]) == len(ref["mems"]):
                nbad = int((res["mems"] != ref["mems"]).sum())
            print("shift", shift, "force", hex(force), "trial", trial, "find_mems identical:", ok, "n_mems", len(res["mems"]), len(ref["mems"]), "differing MEMs", nbad, flush=True)
        idx.close()

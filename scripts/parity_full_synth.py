"""Full-size parity on the synthetic pangenome bench workload (GPU box): the device result of the 1 M-read batch of
`bench.py --workload synth` against the CPU oracle, every MEM, run count and position.
python3 -u scripts/parity_full_synth.py [n_reads [base_len]]   (base_len 40000000 = the chr22-scale index, n = 640 M)"""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pgx_ffi as P, pgx_workload as W, oracle_ffi as O

wd = "/tmp/pgx_parity_full"; os.makedirs(wd, exist_ok=True)
text = os.path.join(wd, "synth.txt")
base_len = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
W.synth_pangenome_text(text, base_len=base_len)
ri, tags = W.build_index_from_text(text, wd, "synth")[:2]
seqs = W.load_sequences(text)
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cat, offs = W.sample_reads(seqs, n_reads, 150, seed=42 + 3)
r, t = O.RIndex(ri), O.Tags(tags, O.TAGS_COMPACT)
t0 = time.time()
ref = O.find_mems_batch(r, t, cat, offs, 20, 1, threads=O.lib().orc_max_threads())
print("oracle: %.1f s, %d MEMs, %d positions, %d extensions" % (time.time() - t0, len(ref["mems"]), len(ref["positions"]), ref["n_extensions"]), flush=True)
for name, force in (("automatic", 0), ("dense2 + pairs", P.MODE_IMAGE_PAIRS), ("dense2", P.MODE_IMAGE_DENSE2), ("dense", P.MODE_IMAGE_DENSE), ("run-length", P.MODE_IMAGE_RL)):
    idx = P.Index(ri, tags, mode=P.MODE_COMPAT | force)
    res = idx.find_mems(cat, offs, 20, 1, tags=True)
    ok = (np.array_equal(res["mem_offsets"], ref["mem_offsets"]) and res["mems"].tobytes() == ref["mems"].tobytes()
          and res["n_extensions"] == ref["n_extensions"] and np.array_equal(res["tag_run_counts"], ref["tag_run_counts"])
          and np.array_equal(res["pos_offsets"], ref["pos_offsets"]) and np.array_equal(res["positions"], ref["positions"])
          and res["n_tag_overflow"] == ref["n_tag_overflow"])
    print("%s image (kind %d, pairs %d): %s (n = %d, %d reads)" % (name, idx.info().image_kind, idx.info().image_pairs, "bit-identical to the oracle" if ok else "MISMATCH", idx.info().bwt_size, n_reads), flush=True)
    assert ok
    idx.close()

import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pgx_workload as W, pgx_ffi as P, oracle_ffi as O
from image_emu import Consts
wd = tempfile.mkdtemp()
text = os.path.join(wd, "w.txt")
W.synth_pangenome_text(text, base_len=30_000, n_hap=3, seed=77, snp=0.01, indel=0.001, n_runs=3, n_run_len=(20, 400))
ri, tags, _ = W.build_index_from_text(text, wd, "w")
o = O.RIndex(ri)
for shift, wide in (("22", 0), ("22", P.MODE_IMAGE_WIDE), ("2", P.MODE_IMAGE_WIDE)):
    os.environ["PGX_SB_SHIFT"] = shift
    for force in (P.MODE_IMAGE_DENSE2, P.MODE_IMAGE_PAIRS, P.MODE_IMAGE_DENSE, P.MODE_IMAGE_RL):
        if wide and force in (P.MODE_IMAGE_DENSE, P.MODE_IMAGE_RL):
            continue
        idx = P.Index(ri, None, mode=force | wide)
        c = Consts(idx.image_view(6))
        pos = np.arange(0, o.n + 2, dtype=np.uint64)
        got = idx.rank_batch(pos, true_codes=True)
        bad = 0
        for p in range(0, o.n + 1, 13):
            exp = o.rank6_true(p)
            if list(got[p]) != exp:
                bad += 1
                if bad <= 5:
                    print("shift", shift, "force", hex(force), "p", p, "blk", p // 384, "sb", (p // 384) >> c.d2_sb_shift, "rel", p % 384, "got", [int(v) for v in got[p]], "exp", exp)
        got2 = idx.rank_batch(pos, true_codes=True)
        for w in (0, 15, 22) + ((20, 23) if force == P.MODE_IMAGE_PAIRS else ()):
            hv, dv = idx.image_view(w), idx.device_view(w)
            neq = np.flatnonzero(hv.view(np.uint8) != dv.view(np.uint8))
            print("  view", w, "bytes", hv.nbytes, "device differs at", len(neq), "bytes", (int(neq[0]), int(neq[-1])) if len(neq) else "")
        print("second call identical:", bool(np.array_equal(got, got2)), "wide", wide)
        print("shift", shift, "d2_sb_shift", c.d2_sb_shift, "n_sb2", c.n_sb2, "force", hex(force), "mismatching positions:", bad, flush=True)
        idx.close()

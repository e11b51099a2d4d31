#!/usr/bin/env python3
"""One-shot probe of the round-3 "stale counts" anomaly (gpurun_out/r3_t2.log): the test-only rank kernel in the
shape it had until commit 66308c2 (one thread loops over the six slots; pos / 384 by a 64-bit mulhi) against today's (one slot per thread; shift +
32-bit multiply), each change on its own, on the index and image of tests/test_wide_image.py::wide_case (wide dense2, PGX_SB_SHIFT=2).  Every
variant runs in a process of its own (the original failure was the first GPU work of its process), three calls each; for every wrong value the
report says which slot, which block / sub-block / workgroup, and whether the value is the right answer for ANOTHER position (a stale or misplaced
result) or for the same position in another slot.  Run once; the output is the evidence (profiles/r04_anomaly_probe.txt)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(variant, wd):
    import numpy as np

    import oracle_ffi as O
    import pgx_ffi as P

    ri = os.path.join(wd, "wide_case.ri")
    o = O.RIndex(ri)
    n = o.n
    exp = np.array([o.rank6_true(min(p, n)) for p in range(n + 2)], dtype=np.uint64)
    os.environ["PGX_SB_SHIFT"] = "2"
    if variant != "current":
        os.environ["PGX_RANK_PROBE"] = variant
    for force in (P.MODE_IMAGE_PAIRS, P.MODE_IMAGE_DENSE2):
        idx = P.Index(ri, None, mode=P.MODE_COMPAT | force | P.MODE_IMAGE_WIDE)
        pos = np.arange(0, n + 2, dtype=np.uint64)
        for call in range(3):
            got = idx.rank_batch(pos, true_codes=True)
            bad = np.argwhere(got != exp)
            line = "variant %-10s force %#x call %d: %d wrong values at %d positions" % (variant, force, call, len(bad), len(np.unique(bad[:, 0])))
            if len(bad):
                slots = np.bincount(bad[:, 1], minlength=6)
                p = bad[:, 0]
                wg = p // 256 if variant.startswith("loop") else (p * 6 + bad[:, 1]) // 256
                line += "; by slot %s; workgroups %d..%d (%d distinct, mod 8: %s); sub-blocks %s" % (
                    slots.tolist(), wg.min(), wg.max(), len(np.unique(wg)), np.bincount(np.unique(wg) % 8, minlength=8).tolist(), np.bincount((p % 384) >> 7, minlength=3).tolist())
                # is a wrong value the right answer somewhere else?
                other_pos = other_slot = 0
                for q, sl in bad[:200]:
                    v = got[q, sl]
                    if (exp[:, sl] == v).any():
                        other_pos += 1
                    if (exp[q] == v).any():
                        other_slot += 1
                q, sl = bad[0]
                line += "; of the first %d: %d equal the right value of another position (same slot), %d of another slot (same position); first: p=%d slot=%d got=%d exp=%d" % (
                    min(len(bad), 200), other_pos, other_slot, q, sl, int(got[q, sl]), int(exp[q, sl]))
            print(line, flush=True)
        idx.close()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3])
    import pgx_workload as W

    wd = tempfile.mkdtemp(prefix="pgx_probe_")
    text = os.path.join(wd, "wide_case.txt")
    W.synth_pangenome_text(text, base_len=30_000, n_hap=3, seed=77, snp=0.01, indel=0.001, n_runs=3, n_run_len=(20, 400))
    W.build_index_from_text(text, wd, "wide_case", with_tags=True)
    # the old shape first, as the failing test met it (first GPU work of a fresh process); this process never touches the GPU
    for variant in ("loop_mulhi", "loop", "mulhi", "current", "loop_mulhi"):
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", variant, wd])
        if rc:
            print("variant %s: child exit code %d" % (variant, rc), flush=True)


if __name__ == "__main__":
    main()

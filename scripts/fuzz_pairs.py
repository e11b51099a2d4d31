"""Randomised parity of the two-step PAIRS kernel (GPU box): python3 -u scripts/fuzz_pairs.py [first_seed [n_seeds]]
The generator of tests/test_gpu_pairs.py::test_pairs_kernel_on_random_small_indexes over many more seeds; stops at the first difference."""
import os, sys, types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pangenome-index_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_pairs as T


class Env:  # the two monkeypatch calls the test makes
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=True): os.environ.pop(k, None)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
wd = "/tmp/pgx_fuzz_pairs"; os.makedirs(wd, exist_ok=True)
for seed in range(first, first + count):
    T._random_index_case(wd, Env(), seed, T.P.MODE_IMAGE_PAIRS)  # dense2 + pairs, forced, shallow seed table
    T._random_index_case(wd, Env(), seed, 0)                     # the automatic layout (LDS + seed / end tables, or dense + pairs)
    print("seed %d ok" % seed, flush=True)
print("all %d seeds bit-identical to the oracle" % count)

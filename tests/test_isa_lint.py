"""The LCE variants of pgx_find_mems_pairs_kernel load their lines with asm statements and wait for them by hand (pgx_kernels.hip, declaration of `row`):
right only while the compiler puts nothing that touches the loaded registers between such a load and the wait behind it.  scripts/isa_lint.py checks the
device assembly for that; here on the assembly of the sources as they are (one hipcc --save-temps of pgx_kernels.hip, ~20 s)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_between_an_asm_load_and_its_wait(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "pangenome-index_amd", "csrc", "pgx_kernels.hip")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "pangenome-index_amd", "csrc"), "-c", src, "-o", str(tmp_path / "k.o"), "--save-temps"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert len(asm) == 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_lint.py"), str(tmp_path / asm[0])], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("asm loads") == 2 and "8 asm loads" in r.stdout, r.stdout  # the two LCE variants (block every 64 / every 96 positions)

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import isa_lint

    # the check itself: a copy of a loaded register in front of the wait is reported, one behind it is not
    bad, n = isa_lint.lint(["\t;;#ASMSTART\n", "\tglobal_load_dwordx4 v[12:15], v[6:7], off\n", "\t;;#ASMEND\n", "\tv_mov_b32_e32 v1, v13\n",
                            "\t;;#ASMSTART\n", "\ts_waitcnt vmcnt(0)\n", "\t;;#ASMEND\n", "\tv_mov_b32_e32 v2, v13\n"])
    assert n == 1 and len(bad) == 1 and bad[0][2] == [13]

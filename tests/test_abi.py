"""CPU tier: libpgx.so loads and exports every symbol include/pgx.h declares; without a GPU every
compute entry point fails loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pgx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built):
    names = _declared()
    assert len(names) >= 20 and "pgx_find_mems_batch" in names and "pgx_batch_run" in names
    L = ctypes.CDLL(P.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "missing export: " + n
    import __graft_entry__ as G

    assert L.pgx_abi_version() == G.header_abi_version()


def test_driver_entry_build(built):
    """__graft_entry__.build() is what the driver runs on the CPU box every round: it must
    succeed at HEAD (a header bump once left a stale literal in it)."""
    import __graft_entry__ as G

    G.build()


def test_kernels_are_gfx950_code_objects(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", P.LIB_PATH], capture_output=True, text=True)
    so = open(P.LIB_PATH, "rb").read()
    assert b"gfx950" in so and b"pgx_find_mems_kernel" in so
    assert b"gfx942" not in so and b"sm_" not in so[:0]  # single-target build
    assert out.returncode == 0


def test_no_oracle_linked_into_product(built):
    so = open(P.LIB_PATH, "rb").read()
    assert b"orc_find_all_mems" not in so and b"pgx_oracle" not in so
    for root, _, files in os.walk(os.path.join(ROOT, "pangenome-index_amd")):
        for f in files:
            if f.endswith((".cpp", ".hip", ".h", ".hpp", ".py")):
                txt = open(os.path.join(root, f), errors="replace").read()
                assert "oracle_ffi" not in txt and "pgx_oracle" not in txt, f


def _no_gpu():
    try:
        return P.device_count() == 0
    except P.PgxError as e:
        return e.code == P.ERR_NO_DEVICE


def test_compute_fails_loudly_without_device(built):
    if not _no_gpu():
        pytest.skip("a GPU is present")
    idx = P.Index(os.path.join(O.GOLDEN, "bidirectional_test", "xy.ri"))
    cat, offs = O.pack_reads(["ACGTACGTACGT"])
    for fn in (lambda: idx.find_mems(cat, offs, 5, 1), lambda: idx.rank_batch(np.array([1], dtype=np.uint64)),
               lambda: idx.to_device(0), lambda: P.device_name(0)):
        with pytest.raises(P.PgxError) as e:
            fn()
        assert e.value.code == P.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_argument_errors(built):
    L = P.lib()
    h = ctypes.c_void_p()
    assert L.pgx_index_open(None, None, 0, 0, ctypes.byref(h)) == P.ERR_ARG
    assert L.pgx_index_open(b"x", None, 0, 7, ctypes.byref(h)) == P.ERR_ARG
    assert b"bad mode" in L.pgx_last_error()
    info = P.IndexInfo()
    assert L.pgx_index_info_get(None, ctypes.byref(info)) == P.ERR_ARG
    L.pgx_index_close(None)
    L.pgx_batch_free(None)

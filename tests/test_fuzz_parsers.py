"""Malformed index files must be rejected with a status (or load, if the damage is harmless), never crash the process:
truncations and random byte flips of the reference's own fixtures through pgx_index_open_memory and the oracle loader."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P

BT = os.path.join(O.GOLDEN, "bidirectional_test")


def _open_memory(ri_bytes, tag_bytes, fmt=P.TAGS_AUTO, mode=P.MODE_COMPAT):
    L = P.lib()
    L.pgx_index_open_memory.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    st = L.pgx_index_open_memory(ri_bytes, len(ri_bytes) if ri_bytes else 0, tag_bytes, len(tag_bytes) if tag_bytes else 0, fmt, mode, C.byref(h))
    if st == P.OK:
        inf = P.IndexInfo()
        assert L.pgx_index_info_get(h, C.byref(inf)) == P.OK
        L.pgx_index_close(h)
    else:
        assert st in (P.ERR_FORMAT, P.ERR_UNSUPPORTED, P.ERR_ARG, P.ERR_NOMEM), st
        assert L.pgx_last_error()
    return st


@pytest.mark.parametrize("name", ["bidirectional_test/xy.ri", "two_contig_graph/xy.ri"])
def test_truncated_and_corrupted_ri(built, workdir, name):
    data = open(os.path.join(O.GOLDEN, name), "rb").read()
    rng = np.random.default_rng(123)
    assert _open_memory(data, None) == P.OK
    n_rejected = 0
    for cut in sorted(set(int(v) for v in rng.integers(0, len(data), 120)) | {0, 1, 7, 8, 24, len(data) - 1}):
        n_rejected += _open_memory(data[:cut], None) != P.OK
    assert n_rejected >= 100  # a truncated file is almost always detected (the block stream has no trailer)
    for _ in range(300):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        _open_memory(bytes(b), None)
    # the oracle loader on the same damage (file based)
    p = os.path.join(workdir, "fuzz.ri")
    for cut in (0, 10, 100, len(data) // 2, len(data) - 3):
        open(p, "wb").write(data[:cut])
        with pytest.raises(RuntimeError):
            O.RIndex(p)


@pytest.mark.parametrize("name,fmt", [("xy_bidirectional_compressed.tags", P.TAGS_AUTO), ("xy_bidirectional_compressed.tags", P.TAGS_BYTECODE)])
def test_truncated_and_corrupted_tags(built, name, fmt):
    ri = open(os.path.join(BT, "xy.ri"), "rb").read()
    data = open(os.path.join(BT, name), "rb").read()
    rng = np.random.default_rng(321)
    assert _open_memory(ri, data, fmt) == P.OK
    for cut in sorted(set(int(v) for v in rng.integers(0, len(data), 80)) | {0, 1, 8, len(data) - 1}):
        _open_memory(ri, data[:cut], fmt)
    for _ in range(200):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        _open_memory(ri, bytes(b), fmt)


def test_corrupted_built_encoded_index(built, x_index):
    data = open(x_index[0], "rb").read()
    tags = open(x_index[1], "rb").read()
    rng = np.random.default_rng(7)
    for _ in range(300):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 5))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        for mode in (P.MODE_COMPAT, P.MODE_STRICT | P.MODE_IMAGE_RL):
            _open_memory(bytes(b), None, mode=mode)
    for _ in range(150):
        t = bytearray(tags)
        t[int(rng.integers(0, len(t)))] = int(rng.integers(0, 256))
        _open_memory(data, bytes(t))

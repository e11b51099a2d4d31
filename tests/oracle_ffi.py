"""ctypes binding of oracle/libpgx_oracle.so -- the CPU oracle (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")

MODE_COMPAT, MODE_STRICT = 0, 1
TAGS_BYTECODE, TAGS_COMPACT = 1, 2


class Mem(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("bwt_start", C.c_uint64), ("size", C.c_int64)]


class BiInt(C.Structure):
    _fields_ = [("forward", C.c_uint64), ("reverse", C.c_uint64), ("size", C.c_int64)]


class BatchResult(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64),
        ("mem_offsets", C.POINTER(C.c_uint64)),
        ("mems", C.POINTER(Mem)),
        ("tag_run_counts", C.POINTER(C.c_uint64)),
        ("pos_offsets", C.POINTER(C.c_uint64)),
        ("positions", C.POINTER(C.c_uint64)),
        ("n_extensions", C.c_uint64),
        ("n_tag_overflow", C.c_uint64),
        ("seconds_mems", C.c_double),
        ("seconds_tags", C.c_double),
    ]


MEM_DTYPE = np.dtype([("start", "<u8"), ("end", "<u8"), ("bwt_start", "<u8"), ("size", "<i8")])

_lib = None


NO_POSITION = 0xFFFFFFFFFFFFFFFF


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "libpgx_oracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    u64, p = C.c_uint64, C.c_void_p
    L.orc_last_error.restype = C.c_char_p
    L.orc_ri_load.restype = p
    L.orc_ri_load.argtypes = [C.c_char_p]
    L.orc_ri_free.argtypes = [p]
    for name in ("bwt_size", "sigma", "n_blocks", "n_block_starts", "max_length", "samples_size",
                 "last_ones", "last_size", "encoded_stream_bytes", "file_bytes_consumed"):
        f = getattr(L, "orc_ri_" + name)
        f.restype, f.argtypes = u64, [p]
    for name in ("C", "block_start", "sample", "last_select", "last_to_run", "block_nruns"):
        f = getattr(L, "orc_ri_" + name)
        f.restype, f.argtypes = u64, [p, u64]
    L.orc_ri_sym_map.restype, L.orc_ri_sym_map.argtypes = C.c_uint8, [p, u64]
    L.orc_ri_is_encoded.restype, L.orc_ri_is_encoded.argtypes = C.c_int, [p]
    L.orc_ri_has_N.restype, L.orc_ri_has_N.argtypes = C.c_int, [p]
    L.orc_ri_block_run.argtypes = [p, u64, u64, C.POINTER(u64), C.POINTER(u64)]
    L.orc_ri_block_cum.restype, L.orc_ri_block_cum.argtypes = u64, [p, u64, u64]
    L.orc_rank_at_cached.argtypes = [p, u64, C.POINTER(u64)]
    L.orc_rank6_true.argtypes = [p, u64, C.POINTER(u64)]
    L.orc_backward_extend.restype = BiInt
    L.orc_backward_extend.argtypes = [p, C.c_int, BiInt, C.c_uint8]
    L.orc_forward_extend.restype = BiInt
    L.orc_forward_extend.argtypes = [p, C.c_int, BiInt, C.c_uint8]
    L.orc_find_all_mems.restype = u64
    L.orc_find_all_mems.argtypes = [p, C.c_int, C.c_char_p, u64, u64, u64, C.POINTER(Mem), u64, C.POINTER(u64)]
    L.orc_find_mems_function.restype = u64
    L.orc_find_mems_function.argtypes = [p, C.c_int, C.c_char_p, u64, u64, u64, u64, C.POINTER(Mem), C.POINTER(C.c_int), C.POINTER(u64)]
    L.orc_count.argtypes = [p, C.c_int, C.c_char_p, u64, C.POINTER(u64), C.POINTER(u64)]
    L.orc_LF.argtypes = [p, C.c_int, C.c_uint8, C.POINTER(u64), C.POINTER(u64)]
    L.orc_locate_first.restype, L.orc_locate_first.argtypes = u64, [p]
    L.orc_locate_next.restype, L.orc_locate_next.argtypes = u64, [p, u64]
    L.orc_seq_id.restype, L.orc_seq_id.argtypes = u64, [p, u64]
    L.orc_seq_offset.restype, L.orc_seq_offset.argtypes = u64, [p, u64]
    L.orc_decompress_sa.argtypes = [p, C.c_void_p]
    L.orc_locate_sa.restype, L.orc_locate_sa.argtypes = u64, [p, C.c_int, u64, u64, C.c_void_p]
    L.orc_tags_load.restype, L.orc_tags_load.argtypes = p, [C.c_char_p, C.c_int]
    L.orc_tags_free.argtypes = [p]
    for name in ("n_runs", "bwt_intervals_size", "n_items", "n_starts", "file_bytes_consumed"):
        f = getattr(L, "orc_tags_" + name)
        f.restype, f.argtypes = u64, [p]
    for name in ("start", "interval", "item"):
        f = getattr(L, "orc_tags_" + name)
        f.restype, f.argtypes = u64, [p, u64]
    L.orc_tags_query.restype = u64
    L.orc_tags_query.argtypes = [p, u64, u64, C.POINTER(u64), C.POINTER(u64), u64, C.POINTER(C.c_int)]
    L.orc_find_mems_batch.restype = C.POINTER(BatchResult)
    L.orc_find_mems_batch.argtypes = [p, p, C.c_int, C.c_void_p, C.c_void_p, u64, u64, u64, C.c_int]
    L.orc_batch_free.argtypes = [C.POINTER(BatchResult)]
    L.orc_max_threads.restype = C.c_int
    _lib = L
    return L


def pack_reads(reads):
    """list of bytes/str -> (concat uint8 array, offsets uint64 array)."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    cat = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return cat, offs


class RIndex:
    def __init__(self, path):
        self.L = lib()
        self.h = self.L.orc_ri_load(path.encode())
        if not self.h:
            raise RuntimeError(self.L.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_ri_free(self.h)
            self.h = None

    n = property(lambda s: s.L.orc_ri_bwt_size(s.h))
    sigma = property(lambda s: s.L.orc_ri_sigma(s.h))
    encoded = property(lambda s: bool(s.L.orc_ri_is_encoded(s.h)))
    has_N = property(lambda s: bool(s.L.orc_ri_has_N(s.h)))
    n_blocks = property(lambda s: s.L.orc_ri_n_blocks(s.h))
    n_block_starts = property(lambda s: s.L.orc_ri_n_block_starts(s.h))

    def C_array(self):
        return [self.L.orc_ri_C(self.h, i) for i in range(self.sigma)]

    def sym_map(self):
        return [self.L.orc_ri_sym_map(self.h, c) for c in range(256)]

    def block_starts(self):
        return [self.L.orc_ri_block_start(self.h, i) for i in range(self.n_block_starts)]

    def rank_at_cached(self, pos):
        out = (C.c_uint64 * 8)()
        self.L.orc_rank_at_cached(self.h, pos, out)
        return list(out[: self.sigma])

    def rank6_true(self, pos):
        out = (C.c_uint64 * 6)()
        self.L.orc_rank6_true(self.h, pos, out)
        return list(out)

    def full(self):
        return (0, 0, self.n)

    def bwd(self, tri, a, mode=MODE_COMPAT):
        a = ord(a) if isinstance(a, str) else a
        o = self.L.orc_backward_extend(self.h, mode, BiInt(*tri), a)
        return (o.forward, o.reverse, o.size)

    def fwd(self, tri, a, mode=MODE_COMPAT):
        a = ord(a) if isinstance(a, str) else a
        o = self.L.orc_forward_extend(self.h, mode, BiInt(*tri), a)
        return (o.forward, o.reverse, o.size)

    def bwd_pattern(self, pat, mode=MODE_COMPAT):
        """extend right-to-left from the full interval (SURVEY 8c convention)."""
        tri = self.full()
        for ch in reversed(pat):
            tri = self.bwd(tri, ch, mode)
        return tri

    def count(self, read, mode=MODE_COMPAT):
        b = read.encode() if isinstance(read, str) else bytes(read)
        lo, hi = C.c_uint64(0), C.c_uint64(0)
        self.L.orc_count(self.h, mode, b, len(b), C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def LF(self, rng, sym, mode=MODE_COMPAT):
        """FastLocate::LF / LF_encoded on the inclusive range rng = (first, second)"""
        sym = ord(sym) if isinstance(sym, str) else sym
        lo, hi = C.c_uint64(rng[0]), C.c_uint64(rng[1])
        self.L.orc_LF(self.h, mode, sym, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    max_length = property(lambda s: s.L.orc_ri_max_length(s.h))

    def locate_first(self):
        return self.L.orc_locate_first(self.h)

    def locate_next(self, prev):
        return self.L.orc_locate_next(self.h, prev)

    def decompress_sa(self):
        out = np.zeros(self.n, dtype=np.uint64)
        self.L.orc_decompress_sa(self.h, out.ctypes.data)
        return out

    def decompress_da(self):
        return self.decompress_sa() // np.uint64(self.max_length)

    def locate_sa(self, first, last, mode=MODE_COMPAT):
        """SA values of BWT[first..last] in BWT order; None when the reference's literal scan is undefined."""
        if last < first:
            return np.zeros(0, dtype=np.uint64)
        out = np.zeros(last - first + 1, dtype=np.uint64)
        k = self.L.orc_locate_sa(self.h, mode, first, last, out.ctypes.data)
        if k == NO_POSITION:
            return None
        return out[:k]

    def locate(self, first, last, mode=MODE_COMPAT):
        """FastLocate::locate / locate_encoded: sorted unique sequence ids."""
        sa = self.locate_sa(first, last, mode)
        if sa is None:
            return None
        return np.unique(sa // np.uint64(self.max_length))

    def find_mems_function(self, read, min_len, min_occ, x, mode=MODE_COMPAT):
        """one call at start x -> (next_x, mem tuple or None, extensions)"""
        b = read.encode() if isinstance(read, str) else bytes(read)
        m, has, ne = Mem(), C.c_int(0), C.c_uint64(0)
        nx = self.L.orc_find_mems_function(self.h, mode, b, len(b), min_len, min_occ, x, C.byref(m), C.byref(has), C.byref(ne))
        return nx, ((m.start, m.end, m.bwt_start, m.size) if has.value else None), ne.value

    def find_all_mems(self, read, min_len, min_occ, mode=MODE_COMPAT, with_ext=False):
        b = read.encode() if isinstance(read, str) else bytes(read)
        cap = max(len(b) + 1, 1)
        out = (Mem * cap)()
        ne = C.c_uint64(0)
        n = self.L.orc_find_all_mems(self.h, mode, b, len(b), min_len, min_occ, out, cap, C.byref(ne))
        mems = [(m.start, m.end, m.bwt_start, m.size) for m in out[:n]]
        return (mems, ne.value) if with_ext else mems


class Tags:
    def __init__(self, path, fmt):
        self.L = lib()
        self.h = self.L.orc_tags_load(path.encode(), fmt)
        if not self.h:
            raise RuntimeError(self.L.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_tags_free(self.h)
            self.h = None

    n_runs = property(lambda s: s.L.orc_tags_n_runs(s.h))
    n_items = property(lambda s: s.L.orc_tags_n_items(s.h))
    n_starts = property(lambda s: s.L.orc_tags_n_starts(s.h))

    def query(self, start, end):
        rn = C.c_uint64(0)
        over = C.c_int(0)
        cap = 16
        while True:
            out = (C.c_uint64 * cap)()
            u = self.L.orc_tags_query(self.h, start, end, C.byref(rn), out, cap, C.byref(over))
            if u <= cap:
                return rn.value, list(out[:u]), bool(over.value)
            cap = u


def find_mems_batch(ri, tags, reads_cat, offsets, min_len, min_occ, mode=MODE_COMPAT, threads=1):
    """Returns dict of numpy arrays (copied) + counters."""
    L = lib()
    reads_cat = np.ascontiguousarray(reads_cat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    res = L.orc_find_mems_batch(ri.h, tags.h if tags is not None else None, mode,
                                reads_cat.ctypes.data, offsets.ctypes.data, n, min_len, min_occ, threads)
    r = res.contents
    mem_offsets = np.ctypeslib.as_array(r.mem_offsets, shape=(n + 1,)).copy()
    nm = int(mem_offsets[-1])
    if nm:
        mems = np.frombuffer(C.string_at(r.mems, nm * 32), dtype=MEM_DTYPE).copy()
    else:
        mems = np.zeros(0, dtype=MEM_DTYPE)
    out = dict(mem_offsets=mem_offsets, mems=mems, n_extensions=int(r.n_extensions),
               seconds_mems=float(r.seconds_mems), seconds_tags=float(r.seconds_tags),
               n_tag_overflow=int(r.n_tag_overflow))
    if tags is not None:
        out["tag_run_counts"] = (np.ctypeslib.as_array(r.tag_run_counts, shape=(nm,)).copy() if nm else np.zeros(0, np.uint64))
        po = np.ctypeslib.as_array(r.pos_offsets, shape=(nm + 1,)).copy()
        out["pos_offsets"] = po
        npz = int(po[-1])
        out["positions"] = (np.ctypeslib.as_array(r.positions, shape=(npz,)).copy() if npz else np.zeros(0, np.uint64))
    L.orc_batch_free(res)
    return out

"""ctypes binding of libpgx.so (include/pgx.h) -- the test / bench harness side of the C ABI.

This module holds no algorithm: it only marshals numpy arrays across the C ABI.  The library has no
CPU fallback; every compute call needs a gfx950 device and raises PgxError otherwise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("PGX_LIB") or os.path.join(PKG_DIR, "libpgx.so")  # PGX_LIB: sanitizer build (scripts/asan_host.sh)

OK, ERR_IO, ERR_FORMAT, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_ARG, ERR_NOMEM = range(8)
MODE_COMPAT, MODE_STRICT = 0, 1
MODE_IMAGE_RL, MODE_IMAGE_DENSE, MODE_IMAGE_DENSE2, MODE_IMAGE_PAIRS = 0x100, 0x200, 0x400, 0x800  # or-ed into mode: force the layout of the device rank image
MODE_IMAGE_WIDE = 0x1000  # modifier: the 64-bit form of dense2 / pairs
IMAGE_RL, IMAGE_DENSE, IMAGE_DENSE2 = 0, 1, 2
TAGS_AUTO, TAGS_BYTECODE, TAGS_COMPACT = 0, 1, 2
RUN_TAGS, RUN_TIMING = 1, 2

MEM_DTYPE = np.dtype([("start", "<u8"), ("end", "<u8"), ("bwt_start", "<u8"), ("size", "<i8")])
BIINT_DTYPE = np.dtype([("forward", "<u8"), ("reverse", "<u8"), ("size", "<i8")])

u64, u32, p = C.c_uint64, C.c_uint32, C.c_void_p


class IndexInfo(C.Structure):
    _fields_ = [
        ("bwt_size", u64), ("sigma", u64), ("n_sequences", u64), ("n_ref_blocks", u64), ("n_runs", u64),
        ("n_dev_blocks", u64), ("dir_entries", u64), ("dir_shift", u32), ("is_encoded", u32), ("has_N", u32),
        ("mode", u32), ("has_tags", u32), ("tag_format", u32), ("n_tag_runs", u64), ("tag_dir_entries", u64),
        ("tag_dir_shift", u32), ("image_in_lds", u32), ("image_bytes", u64), ("tag_image_bytes", u64),
        ("ref_block_mean_bytes", C.c_double), ("max_length", u64), ("n_samples", u64), ("image_kind", u32), ("image_pairs", u32),
        ("image_wide", u32), ("pairs_stride", u32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("n_reads", u64), ("n_mems", u64), ("mem_offsets", C.POINTER(u64)), ("mems", p),
        ("tag_run_counts", C.POINTER(u64)), ("pos_offsets", C.POINTER(u64)), ("positions", C.POINTER(u64)),
        ("n_positions", u64), ("n_extensions", u64), ("n_tag_overflow", u64),
    ]


class DeviceResult(C.Structure):
    _fields_ = [("n_reads", u64), ("n_mems", u64), ("n_positions", u64), ("mem_offsets", p), ("mems", p),
                ("tag_run_counts", p), ("pos_offsets", p), ("positions", p)]


class DeviceArray:
    """a device buffer owned by a batch, exposed through __cuda_array_interface__ (zero copy: torch.as_tensor(a, device="cuda"));
    64-bit unsigned values are presented as int64"""

    def __init__(self, ptr, shape, owner):
        self.owner = owner  # keeps the batch alive
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<i8", "data": (int(ptr or 0), False), "version": 2, "strides": None}


class ExchangeResult(C.Structure):
    _fields_ = [("n_reads", u64), ("n_mems", u64), ("mem_offsets", p), ("mems", p), ("shard_of_mem", p)]


class Timing(C.Structure):
    _fields_ = [
        ("ms_find_mems", C.c_float), ("ms_compact", C.c_float), ("ms_tag_locate", C.c_float),
        ("ms_tag_gather", C.c_float), ("ms_tag_sort", C.c_float), ("ms_total", C.c_float),
        ("find_mems_launches", u32), ("heavy_reads", u32), ("pairs_reads", u32), ("pairs_other_steps", u32),
        ("ms_find_mems_main", C.c_float), ("seed_depth", u32),
        ("main_lines", u64), ("main_seed_loads", u64), ("other_lines", u64), ("other_seed_loads", u64), ("two_step_trips", u64),
        ("ms_per_upload", C.c_float),
    ]


def _u64_array(ptr, n):
    """copy n uint64 from a (possibly NULL) ctypes pointer"""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=np.uint64)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


class PgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pgx status %d: %s" % (code, msg))
        self.code = code


_lib = None


def build():
    """Compile libpgx.so in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    subprocess.run(["make", "-C", PKG_DIR, "-s", "-j4"], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PgxError(ERR_NO_DEVICE, "libpgx.so is not built (run __graft_entry__.build()); there is no fallback")
    L = C.CDLL(LIB_PATH)
    L.pgx_last_error.restype = C.c_char_p
    L.pgx_abi_version.restype = C.c_int
    L.pgx_index_open.argtypes = [C.c_char_p, C.c_char_p, u32, u32, C.POINTER(p)]
    L.pgx_index_info_get.argtypes = [p, C.POINTER(IndexInfo)]
    L.pgx_index_close.argtypes = [p]
    L.pgx_index_close.restype = None
    L.pgx_index_to_device.argtypes = [p, C.c_int]
    L.pgx_index_image_view.argtypes = [p, C.c_int, C.POINTER(p), C.POINTER(u64)]
    L.pgx_build_rindex.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.pgx_build_rlbwt.argtypes = [C.c_char_p, C.c_char_p]
    L.pgx_build_index_from_text.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    L.pgx_write_compact_tags.argtypes = [C.c_char_p, p, p, u64]
    L.pgx_convert_tags.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.pgx_merge_tags.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), u32, p, u64, C.c_int, C.c_char_p]
    L.pgx_gbz_paths.argtypes = [C.c_char_p, C.POINTER(u64), p, p, u64, C.POINTER(u64), C.POINTER(u32)]
    L.pgx_merge_tags_gbz.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u32, C.c_int, C.c_char_p]
    L.pgx_merge_tags_ex.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), u32, p, u64, C.c_int, C.c_char_p, u32]
    L.pgx_merge_tags_gbz_ex.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u32, C.c_int, C.c_char_p, u32]
    L.pgx_rank_batch.argtypes = [p, C.c_int, p, u64, C.c_int, p]
    L.pgx_extend_batch.argtypes = [p, C.c_int, p, p, p, u64, p]
    L.pgx_count_batch.argtypes = [p, C.c_int, p, p, u64, p]
    L.pgx_find_mems_function_batch.argtypes = [p, C.c_int, p, p, u64, p, p, u64, u64, u64, p, p, p, p]
    L.pgx_lf_batch.argtypes = [p, C.c_int, p, p, u64, p]
    L.pgx_tag_query_batch.argtypes = [p, C.c_int, p, p, u64, p, p, p, u64, C.POINTER(u64)]
    L.pgx_locate_batch.argtypes = [p, C.c_int, p, p, u64, u32, p, p, u64]
    L.pgx_locate_next_batch.argtypes = [p, C.c_int, p, u64, p]
    L.pgx_decompress_sa.argtypes = [p, C.c_int, u32, p]
    L.pgx_batch_create.argtypes = [p, C.c_int, p, p, u64, C.POINTER(p)]
    L.pgx_batch_upload.argtypes = [p, p, p, u64]
    L.pgx_batch_upload_packed.argtypes = [p, p, p, u64, p, p, u64]
    L.pgx_pack_reads.argtypes = [p, p, u64, u32, p, p, u64, p, u64, C.POINTER(u64), C.POINTER(u64)]
    L.pgx_host_alloc.argtypes = [C.c_size_t, C.POINTER(p)]
    L.pgx_host_free.argtypes = [p]
    L.pgx_host_free.restype = None
    L.pgx_batch_run.argtypes = [p, u64, u64, u32, p]
    L.pgx_batch_result.argtypes = [p, C.POINTER(Result)]
    L.pgx_batch_counts.argtypes = [p, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.pgx_batch_device_result.argtypes = [p, C.POINTER(DeviceResult)]
    L.pgx_batch_timing.argtypes = [p, C.POINTER(Timing)]
    L.pgx_batch_spec_stats.argtypes = [p, C.POINTER(u32), C.POINTER(u32)]
    L.pgx_batch_free.argtypes = [p]
    L.pgx_batch_free.restype = None
    L.pgx_find_mems_batch.argtypes = [p, C.c_int, p, p, u64, u64, u64, u32, C.POINTER(p), C.POINTER(Result)]
    L.pgx_find_mems_sharded.argtypes = [p, p, u32, p, p, u64, u64, u64, u32, p, p, p]
    L.pgx_comm_unique_id.argtypes = [p]
    L.pgx_comm_init.argtypes = [p, C.c_int, C.c_int, C.c_int, C.POINTER(p)]
    L.pgx_comm_free.argtypes = [p]
    L.pgx_comm_free.restype = None
    L.pgx_exchange_mems.argtypes = [p, p, p, u32, p, u32, C.POINTER(ExchangeResult)]
    L.pgx_exchange_download.argtypes = [p, p, p, p]
    L.pgx_device_count.argtypes = [C.POINTER(C.c_int)]
    L.pgx_device_name.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    _lib = L
    return L


def _check(st):
    if st != OK:
        raise PgxError(st, lib().pgx_last_error().decode(errors="replace"))


def device_count():
    n = C.c_int(0)
    _check(lib().pgx_device_count(C.byref(n)))
    return n.value


def device_name(device=0):
    buf = C.create_string_buffer(256)
    _check(lib().pgx_device_name(device, buf, 256))
    return buf.value.decode()


class _Pinned:
    """owner of one pgx_host_alloc block (freed with the last numpy view on it)"""

    def __init__(self, nbytes):
        self.ptr = p()
        _check(lib().pgx_host_alloc(max(int(nbytes), 16), C.byref(self.ptr)))
        self.nbytes = max(int(nbytes), 16)

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().pgx_host_free(self.ptr)
            self.ptr = None


def pinned_array(n, dtype):
    """numpy array of n items in pinned host memory (pgx_host_alloc): uploads from it and downloads into it run at link speed"""
    dt = np.dtype(dtype)
    own = _Pinned(int(n) * dt.itemsize)
    buf = (C.c_uint8 * own.nbytes).from_address(own.ptr.value)
    buf._pgx_owner = own  # (the array's base is this ctypes object: the block lives as long as any view on it)
    return np.frombuffer(buf, dtype=dt, count=int(n))


def pack_reads(reads_cat, offsets, packed_out, side_ids_out, side_bytes_out, threads=0):
    """pgx_pack_reads: the batch packed to two bits per symbol into packed_out (uint32[(bytes + 15) // 16]) and the reads with a byte outside
    A C G T listed (ids, their bytes concatenated) -> (n_side, n_side_bytes); PgxError(ERR_NOMEM) when the list does not fit"""
    reads_cat = np.ascontiguousarray(reads_cat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    assert packed_out.dtype == np.uint32 and len(packed_out) >= (len(reads_cat) + 15) // 16
    ns, nb = u64(0), u64(0)
    _check(lib().pgx_pack_reads(reads_cat.ctypes.data if len(reads_cat) else None, offsets.ctypes.data, len(offsets) - 1, threads, packed_out.ctypes.data,
                                side_ids_out.ctypes.data, len(side_ids_out), side_bytes_out.ctypes.data, len(side_bytes_out), C.byref(ns), C.byref(nb)))
    return ns.value, nb.value


def build_rindex(rlbwt_path, out_path, encoded=True):
    _check(lib().pgx_build_rindex(rlbwt_path.encode(), out_path.encode(), 1 if encoded else 0))


def build_rlbwt(text_path, out_path):
    _check(lib().pgx_build_rlbwt(text_path.encode(), out_path.encode()))


def build_index_from_text(text_path, out_rlbwt_path, out_ri_path, encoded=True):
    _check(lib().pgx_build_index_from_text(text_path.encode(), out_rlbwt_path.encode() if out_rlbwt_path else None, out_ri_path.encode(),
                                           1 if encoded else 0))


def build_index_from_texts(text_paths, out_rlbwt_path, out_ri_path, encoded=True):
    arr = (C.c_char_p * len(text_paths))(*[t.encode() for t in text_paths])
    _check(lib().pgx_build_index_from_texts(arr, len(text_paths), out_rlbwt_path.encode() if out_rlbwt_path else None, out_ri_path.encode(),
                                            1 if encoded else 0))


def write_compact_tags(out_path, values, lengths):
    v = np.ascontiguousarray(values, dtype=np.uint64)
    l = np.ascontiguousarray(lengths, dtype=np.uint64)
    assert len(v) == len(l)
    _check(lib().pgx_write_compact_tags(out_path.encode(), v.ctypes.data, l.ctypes.data, len(v)))


def convert_tags(in_path, out_path, compact=False):
    _check(lib().pgx_convert_tags(in_path.encode(), out_path.encode(), 1 if compact else 0))


_VIEW_DTYPES = {0: np.uint8, 1: np.uint64, 2: np.uint64, 3: np.uint64, 4: np.uint64, 5: np.uint32, 6: np.uint8, 7: np.uint16,
                8: np.uint64, 9: np.uint64, 10: np.uint32, 11: np.uint64, 12: np.uint64, 13: np.uint32, 14: np.uint8, 15: np.uint32,
                16: np.uint64, 17: np.uint64, 18: np.uint64, 19: np.uint32, 20: np.uint32, 21: np.uint32, 22: np.uint64, 23: np.uint64}
LOCATE_SEQ_IDS, LOCATE_UNIQUE = 1, 2
NO_POSITION = 0xFFFFFFFFFFFFFFFF


MERGE_REFERENCE_RUNS = 0x1


def merge_tags(ri_path, tag_paths, seq_to_file, out_path, device=0, flags=0):
    """merge_tags: per-chromosome tag streams (algorithm format) -> whole-genome sdsl-compact tag array; seq_to_file[s] =
    index into tag_paths of the file that holds the tags of sequence s of the whole-genome r-index.  flags: MERGE_REFERENCE_RUNS
    writes run lengths as the reference does (mod 65 536)"""
    s2f = np.ascontiguousarray(seq_to_file, dtype=np.uint32)
    arr = (C.c_char_p * len(tag_paths))(*[t.encode() for t in tag_paths])
    _check(lib().pgx_merge_tags_ex(ri_path.encode(), arr, len(tag_paths), s2f.ctypes.data, len(s2f), device, out_path.encode(), flags))


def gbz_paths(gbz_path):
    """(first node id per GBWT sequence, its component, max node id, number of components) of a GBZ graph"""
    n, mx, nc = u64(0), u64(0), u32(0)
    _check(lib().pgx_gbz_paths(gbz_path.encode(), C.byref(n), None, None, 0, C.byref(mx), C.byref(nc)))
    first = np.zeros(n.value, dtype=np.uint64)
    comp = np.zeros(n.value, dtype=np.uint32)
    _check(lib().pgx_gbz_paths(gbz_path.encode(), C.byref(n), first.ctypes.data, comp.ctypes.data, n.value, C.byref(mx), C.byref(nc)))
    return first, comp, mx.value, nc.value


def merge_tags_gbz(gbz_path, ri_path, tag_paths, out_path, device=0, flags=0):
    """merge_tags with the reference's inputs: the sequence -> tag file map comes from the graph"""
    arr = (C.c_char_p * len(tag_paths))(*[t.encode() for t in tag_paths])
    _check(lib().pgx_merge_tags_gbz_ex(gbz_path.encode(), ri_path.encode(), arr, len(tag_paths), device, out_path.encode(), flags))


class Index:
    """FastLocate + TagArray behind the C ABI."""

    def __init__(self, ri_path, tags_path=None, tags_format=TAGS_AUTO, mode=MODE_COMPAT):
        self.L = lib()
        self.h = p()
        _check(self.L.pgx_index_open(ri_path.encode(), tags_path.encode() if tags_path else None, tags_format, mode,
                                     C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.L.pgx_index_close(self.h)
            self.h = None

    __del__ = close

    def info(self):
        inf = IndexInfo()
        _check(self.L.pgx_index_info_get(self.h, C.byref(inf)))
        return inf

    def to_device(self, device=0):
        _check(self.L.pgx_index_to_device(self.h, device))

    def image_view(self, which):
        ptr, nbytes = p(), u64(0)
        _check(self.L.pgx_index_image_view(self.h, which, C.byref(ptr), C.byref(nbytes)))
        if nbytes.value == 0:
            return np.zeros(0, dtype=_VIEW_DTYPES[which])
        raw = C.string_at(ptr, nbytes.value)
        return np.frombuffer(raw, dtype=_VIEW_DTYPES[which]).copy()

    def device_view(self, which, device=0):
        """the device copy of image view `which` (0, 15, 20, 22, 23) as bytes"""
        host = self.image_view(which)
        out = np.zeros(host.nbytes, dtype=np.uint8)
        _check(self.L.pgx_index_device_view(self.h, device, which, C.c_void_p(out.ctypes.data), u64(host.nbytes)))
        return out.view(host.dtype)

    def lce_view(self, which, nbytes, device=0):
        """the device's LCE image (which = 30 suffix array, 31 text, 32 line flags, 33 common prefixes) as bytes; zeros where the index has none"""
        out = np.zeros(int(nbytes), dtype=np.uint8)
        _check(self.L.pgx_index_device_view(self.h, device, which, C.c_void_p(out.ctypes.data), u64(int(nbytes))))
        return out

    # ---- primitives -------------------------------------------------------------------------
    def rank_batch(self, pos, true_codes=False, device=0):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros((len(pos), 6), dtype=np.uint64)
        _check(self.L.pgx_rank_batch(self.h, device, pos.ctypes.data, len(pos), 1 if true_codes else 0, out.ctypes.data))
        return out

    def extend_batch(self, intervals, syms, forward, device=0):
        iv = np.ascontiguousarray(intervals, dtype=BIINT_DTYPE)
        sy = np.ascontiguousarray(syms, dtype=np.uint8)
        fw = np.ascontiguousarray(forward, dtype=np.uint8)
        out = np.zeros(len(iv), dtype=BIINT_DTYPE)
        _check(self.L.pgx_extend_batch(self.h, device, iv.ctypes.data, sy.ctypes.data, fw.ctypes.data, len(iv), out.ctypes.data))
        return out

    def count_batch(self, reads_cat, offsets, device=0):
        """FastLocate::count / count_encoded for every read -> uint64[n, 2] (first, second); empty = (1, 0)"""
        reads_cat = np.ascontiguousarray(reads_cat, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.zeros((n, 2), dtype=np.uint64)
        _check(self.L.pgx_count_batch(self.h, device, reads_cat.ctypes.data if len(reads_cat) else None, offsets.ctypes.data, n,
                                      out.ctypes.data))
        return out

    def find_mems_function_batch(self, reads_cat, offsets, read_of, xs, min_len, min_occ, device=0):
        """find_mems_function (algorithm.hpp:653-736) for (read, start) pairs -> (next_x, mems, has_mem, n_ext)"""
        reads_cat = np.ascontiguousarray(reads_cat, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ro = np.ascontiguousarray(read_of, dtype=np.uint64)
        xv = np.ascontiguousarray(xs, dtype=np.uint64)
        n = len(ro)
        nx, ne = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        mems, has = np.zeros(n, dtype=MEM_DTYPE), np.zeros(n, dtype=np.uint8)
        _check(self.L.pgx_find_mems_function_batch(self.h, device, reads_cat.ctypes.data if len(reads_cat) else None, offsets.ctypes.data,
                                                   len(offsets) - 1, ro.ctypes.data, xv.ctypes.data, n, min_len, min_occ, nx.ctypes.data,
                                                   mems.ctypes.data, has.ctypes.data, ne.ctypes.data))
        return nx, mems, has, ne

    def lf_batch(self, ranges, syms, device=0):
        """FastLocate::LF / LF_encoded: uint64[n, 2] inclusive ranges, one symbol each -> uint64[n, 2]; empty = (1, 0)"""
        rg = np.ascontiguousarray(ranges, dtype=np.uint64).reshape(-1, 2)
        sy = np.ascontiguousarray(syms, dtype=np.uint8)
        out = np.zeros((len(rg), 2), dtype=np.uint64)
        _check(self.L.pgx_lf_batch(self.h, device, rg.ctypes.data, sy.ctypes.data, len(rg), out.ctypes.data))
        return out

    def tag_query_batch(self, start, end, device=0):
        st = np.ascontiguousarray(start, dtype=np.uint64)
        en = np.ascontiguousarray(end, dtype=np.uint64)
        n = len(st)
        rn = np.zeros(n, dtype=np.uint64)
        po = np.zeros(n + 1, dtype=np.uint64)
        nover = u64(0)
        _check(self.L.pgx_tag_query_batch(self.h, device, st.ctypes.data, en.ctypes.data, n, rn.ctypes.data, po.ctypes.data,
                                          None, 0, C.byref(nover)))
        P = int(po[-1])
        pos = np.zeros(max(P, 1), dtype=np.uint64)
        _check(self.L.pgx_tag_query_batch(self.h, device, st.ctypes.data, en.ctypes.data, n, rn.ctypes.data, po.ctypes.data,
                                          pos.ctypes.data, len(pos), C.byref(nover)))
        return rn, po, pos[:P], nover.value

    def locate_batch(self, first, last, flags=0, device=0):
        """FastLocate::locate for n BWT ranges -> (offsets[n+1], values); flags: LOCATE_SEQ_IDS | LOCATE_UNIQUE"""
        fi = np.ascontiguousarray(first, dtype=np.uint64)
        la = np.ascontiguousarray(last, dtype=np.uint64)
        n = len(fi)
        off = np.zeros(n + 1, dtype=np.uint64)
        cap = int(np.sum(np.where(la >= fi, la - fi + np.uint64(1), np.uint64(0)))) if n else 0
        vals = np.zeros(max(cap, 1), dtype=np.uint64)
        _check(self.L.pgx_locate_batch(self.h, device, fi.ctypes.data, la.ctypes.data, n, flags, off.ctypes.data, vals.ctypes.data, cap))
        return off, vals[: int(off[-1])]

    def locate_next_batch(self, prev, device=0):
        pv = np.ascontiguousarray(prev, dtype=np.uint64)
        out = np.zeros(len(pv), dtype=np.uint64)
        _check(self.L.pgx_locate_next_batch(self.h, device, pv.ctypes.data, len(pv), out.ctypes.data))
        return out

    def decompress_sa(self, seq_ids=False, device=0):
        out = np.zeros(self.info().bwt_size, dtype=np.uint64)
        _check(self.L.pgx_decompress_sa(self.h, device, LOCATE_SEQ_IDS if seq_ids else 0, out.ctypes.data))
        return out

    # ---- hot path ---------------------------------------------------------------------------
    def batch(self, reads_cat, offsets, device=0):
        return Batch(self, reads_cat, offsets, device)

    def batch_empty(self, device=0):
        """a batch without reads yet (upload / upload_packed fill it)"""
        return Batch(self, np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64), device)

    def find_mems(self, reads_cat, offsets, min_len, min_occ, tags=False, device=0):
        b = Batch(self, reads_cat, offsets, device)
        try:
            b.run(min_len, min_occ, RUN_TAGS if tags else 0)
            return b.result()
        finally:
            b.free()


def _result_dict(r):
    n, m = int(r.n_reads), int(r.n_mems)
    out = dict(n_extensions=int(r.n_extensions), n_tag_overflow=int(r.n_tag_overflow))
    out["mem_offsets"] = _u64_array(r.mem_offsets, n + 1)
    out["mems"] = (np.frombuffer(C.string_at(r.mems, m * 32), dtype=MEM_DTYPE).copy() if m else np.zeros(0, MEM_DTYPE))
    if r.pos_offsets:
        out["tag_run_counts"] = _u64_array(r.tag_run_counts, m)
        out["pos_offsets"] = _u64_array(r.pos_offsets, m + 1)
        out["positions"] = _u64_array(r.positions, int(r.n_positions))
    return out


def find_mems_sharded(index, devices, reads_cat, offsets, min_len, min_occ, tags=False):
    """pgx_find_mems_sharded: contiguous read slices, slice i on devices[i]; returns the per-slice result dicts in slice order"""
    reads_cat = np.ascontiguousarray(reads_cat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    k = len(devices)
    dev = (C.c_int * k)(*devices)
    batches = (p * k)()
    results = (Result * k)()
    first = (u64 * (k + 1))()
    _check(index.L.pgx_find_mems_sharded(index.h, dev, k, reads_cat.ctypes.data if len(reads_cat) else None, offsets.ctypes.data,
                                         len(offsets) - 1, min_len, min_occ, RUN_TAGS if tags else 0, batches, results, first))
    try:
        return [_result_dict(results[i]) for i in range(k)], list(first)
    finally:
        for i in range(k):
            index.L.pgx_batch_free(batches[i])


COMM_ID_BYTES = 128


XCH_META_HEAD = 5
XCH_MAX_OFFSET_BYTES = 4 << 30


def exchange_owner_digest(owner_of_shard):
    L = lib()
    L.pgx_exchange_owner_digest.restype = u64
    own = (u32 * max(len(owner_of_shard), 1))(*owner_of_shard)
    return int(L.pgx_exchange_owner_digest(own, len(owner_of_shard)))


def exchange_plan(world, owner_of_shard, gathered):
    """pgx_exchange_plan (host only, no device): gathered = `world` metadata rows of XCH_META_HEAD + max_local u64 each.
    Returns dict(max_local, slot, rec_base, src_base, n_reads)."""
    L = lib()
    n = len(owner_of_shard)
    own = (u32 * max(n, 1))(*owner_of_shard)
    g = np.ascontiguousarray(gathered, dtype=np.uint64).reshape(-1)
    ml, nr = u32(0), u64(0)
    slot = np.zeros(n, dtype=np.uint32)
    rec_base = np.zeros(world + 1, dtype=np.uint64)
    src_base = np.zeros(n, dtype=np.uint64)
    _check(L.pgx_exchange_plan(u32(world), own, u32(n), C.c_void_p(g.ctypes.data), C.byref(ml), C.c_void_p(slot.ctypes.data),
                               C.c_void_p(rec_base.ctypes.data), C.c_void_p(src_base.ctypes.data), C.byref(nr)))
    return dict(max_local=int(ml.value), slot=slot, rec_base=rec_base, src_base=src_base, n_reads=int(nr.value))


def comm_unique_id():
    """ncclGetUniqueId through libpgx: rank 0 makes it, the other ranks receive the bytes out of band"""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().pgx_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """RCCL communicator of the chromosome-sharded exchange (pgx_comm): one per rank = per process = per GPU"""

    def __init__(self, uid, rank, world, device=0):
        self.L = lib()
        self.c = p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        _check(self.L.pgx_comm_init(buf, rank, world, device, C.byref(self.c)))

    def exchange(self, batches, owner_of_shard, download=True):
        """batches: {shard id: Batch (already run)} of this rank; owner_of_shard[c] = rank holding shard c.
        Returns (n_reads, n_mems) or, with download, (mem_offsets, mems, shard_of_mem) as host arrays."""
        ids = sorted(batches)
        k = len(ids)
        arr = (p * max(k, 1))(*[batches[c].b for c in ids])
        sid = (u32 * max(k, 1))(*ids)
        own = (u32 * len(owner_of_shard))(*owner_of_shard)
        r = ExchangeResult()
        _check(self.L.pgx_exchange_mems(self.c, arr, sid, k, own, len(owner_of_shard), C.byref(r)))
        if not download:
            return int(r.n_reads), int(r.n_mems)
        mo = np.zeros(int(r.n_reads) + 1, dtype=np.uint64)
        mems = np.zeros(int(r.n_mems), dtype=MEM_DTYPE)
        shard = np.zeros(int(r.n_mems), dtype=np.uint32)
        _check(self.L.pgx_exchange_download(self.c, mo.ctypes.data, mems.ctypes.data if len(mems) else None, shard.ctypes.data if len(shard) else None))
        return mo, mems, shard

    def free(self):
        if getattr(self, "c", None):
            self.L.pgx_comm_free(self.c)
            self.c = None

    __del__ = free


class Batch:
    def __init__(self, index, reads_cat, offsets, device=0):
        self.index = index
        self.L = index.L
        reads_cat = np.ascontiguousarray(reads_cat, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n = len(offsets) - 1
        self.b = p()
        _check(self.L.pgx_batch_create(index.h, device, reads_cat.ctypes.data if len(reads_cat) else None,
                                       offsets.ctypes.data, self.n, C.byref(self.b)))

    def upload_packed(self, packed, offsets, side_ids, side_bytes, n_side):
        """pgx_batch_upload_packed: the arrays pack_reads() filled (offsets[0] == 0)"""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        assert packed.dtype == np.uint32 and packed.flags.c_contiguous
        self.n = len(offsets) - 1
        _check(self.L.pgx_batch_upload_packed(self.b, packed.ctypes.data, offsets.ctypes.data, self.n, side_ids.ctypes.data if n_side else None,
                                              side_bytes.ctypes.data if n_side else None, n_side))

    def upload(self, reads_cat, offsets):
        reads_cat = np.ascontiguousarray(reads_cat, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n = len(offsets) - 1
        _check(self.L.pgx_batch_upload(self.b, reads_cat.ctypes.data if len(reads_cat) else None, offsets.ctypes.data, self.n))

    def run(self, min_len, min_occ, flags=0, stream=None):
        _check(self.L.pgx_batch_run(self.b, min_len, min_occ, flags, stream))

    def counts(self):
        a, b_, c = u64(0), u64(0), u64(0)
        _check(self.L.pgx_batch_counts(self.b, C.byref(a), C.byref(b_), C.byref(c)))
        return a.value, b_.value, c.value

    def spec_stats(self):
        """(runs sized speculatively, of which repeated with exact sizes)"""
        a, f = u32(0), u32(0)
        _check(self.L.pgx_batch_spec_stats(self.b, C.byref(a), C.byref(f)))
        return a.value, f.value

    def timing(self):
        t = Timing()
        _check(self.L.pgx_batch_timing(self.b, C.byref(t)))
        return t

    def result(self):
        r = Result()
        _check(self.L.pgx_batch_result(self.b, C.byref(r)))
        n, m = int(r.n_reads), int(r.n_mems)
        out = dict(n_extensions=int(r.n_extensions), n_tag_overflow=int(r.n_tag_overflow))
        out["mem_offsets"] = _u64_array(r.mem_offsets, n + 1)
        out["mems"] = (np.frombuffer(C.string_at(r.mems, m * 32), dtype=MEM_DTYPE).copy() if m else np.zeros(0, MEM_DTYPE))
        if r.pos_offsets:
            out["tag_run_counts"] = _u64_array(r.tag_run_counts, m)
            out["pos_offsets"] = _u64_array(r.pos_offsets, m + 1)
            out["positions"] = _u64_array(r.positions, int(r.n_positions))
        return out

    def result_counts(self):
        """pgx_batch_result without copying the arrays into numpy: the download into the batch's pinned host buffers happens,
        the (n_mems, n_positions) of the result are returned (the PCIe-inclusive timing of bench.py)"""
        r = Result()
        _check(self.L.pgx_batch_result(self.b, C.byref(r)))
        return int(r.n_mems), int(r.n_positions)

    def device_result(self):
        """results of the last run as device arrays (valid until the next run / upload / free): mem_offsets int64[n + 1],
        mems int64[n_mems, 4] (start, end, bwt_start, size), and with tags tag_run_counts / pos_offsets / positions"""
        r = DeviceResult()
        _check(self.L.pgx_batch_device_result(self.b, C.byref(r)))
        n, m = int(r.n_reads), int(r.n_mems)
        out = dict(device=True, n_reads=n, n_mems=m, mem_offsets=DeviceArray(r.mem_offsets, (n + 1,), self),
                   mems=DeviceArray(r.mems, (m, 4), self))
        if r.pos_offsets:
            out["tag_run_counts"] = DeviceArray(r.tag_run_counts, (m,), self)
            out["pos_offsets"] = DeviceArray(r.pos_offsets, (m + 1,), self)
            out["positions"] = DeviceArray(r.positions, (int(r.n_positions),), self)
        return out

    def free(self):
        if getattr(self, "b", None):
            self.L.pgx_batch_free(self.b)
            self.b = None

    __del__ = free

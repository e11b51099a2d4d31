// pgx_merge_kernels.hip -- merge_tags (src/merge_tags.cpp:289-405,640-820) as data-parallel passes.
//
// The reference walks the whole-genome suffix array in BWT order (locateNext chains), asks for every position which
// chromosome its sequence belongs to and pulls "the next tag" from that chromosome's tag stream; the streams are
// consumed strictly in order.  "The next tag of stream f" for BWT position i is simply element rank_f(i) of the
// expanded stream, rank_f(i) = number of earlier non-endmarker positions that belong to file f -- so the walk becomes
//   DA (pgx_locate kernels) -> file id per position -> per file: exclusive scan of the indicator, gather from the
//   expanded stream -> run-length encode (flags, scan, compact).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgx_device.h"

// file id of every BWT position: 255 for the endmarker block [0, n_seq), else seq_to_file[DA[i]]
__global__ void __launch_bounds__(256)
pgx_mt_file_of_kernel(const uint64_t *__restrict__ da, uint64_t n, uint64_t n_seq, const uint32_t *__restrict__ seq_to_file,
                      uint32_t n_files, uint8_t *__restrict__ file_of, unsigned long long *__restrict__ n_bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t f = 255;
    if (i >= n_seq) {
        const uint64_t s = da[i];
        const uint32_t v = s < n_seq ? seq_to_file[s] : 0xFFFFFFFFu;
        if (v < n_files) f = (uint8_t)v;
        else { f = 254; atomicAdd(n_bad, 1ull); } // a sequence without a file: reported by the host
    }
    file_of[i] = f;
}

// expanded stream of one file: run r fills out[start[r], start[r + 1]) with val[r] (run lengths are < 512)
__global__ void __launch_bounds__(256)
pgx_mt_expand_kernel(const uint64_t *__restrict__ start, const uint64_t *__restrict__ val, uint64_t n_runs, uint64_t *__restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_runs) return;
    const uint64_t a = start[r], b = start[r + 1], v = val[r];
    for (uint64_t t = a; t < b; t++) out[t] = v;
}

// positions of file f take element rank[i] of its expanded stream
__global__ void __launch_bounds__(256)
pgx_mt_gather_kernel(const uint8_t *__restrict__ file_of, uint32_t f, const uint64_t *__restrict__ rank, const uint64_t *__restrict__ expanded,
                     uint64_t total, uint64_t n, uint64_t *__restrict__ tags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || file_of[i] != (uint8_t)f) return;
    const uint64_t r = rank[i];
    tags[i] = r < total ? expanded[r] : 0; // the host has checked that the counts agree
}

// run heads of the merged array (the endmarker block [0, n_seq) is one run of value 0)
__global__ void __launch_bounds__(256)
pgx_mt_flags_kernel(const uint64_t *__restrict__ tags, uint64_t n, uint64_t n_seq, uint8_t *__restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t cur = i < n_seq ? 0 : tags[i];
    const uint64_t prev = (i == 0) ? ~0ull : (i - 1 < n_seq ? 0 : tags[i - 1]);
    flags[i] = (i == 0 || cur != prev) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
pgx_mt_compact_kernel(const uint64_t *__restrict__ tags, const uint8_t *__restrict__ flags, const uint64_t *__restrict__ idx, uint64_t n,
                      uint64_t n_seq, uint64_t *__restrict__ out_val, uint64_t *__restrict__ out_start) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    out_val[idx[i]] = i < n_seq ? 0 : tags[i];
    out_start[idx[i]] = i;
}

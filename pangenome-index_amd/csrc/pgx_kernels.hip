// pgx_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the find_mems hot path.
//
//   pgx_find_mems_kernel   one lane = one read; every loop trip performs exactly one FMD extension
//                          (= two rank probes) for every live lane of the 64-wide wavefront.
//                          Replaces find_all_mems / find_mems_function (algorithm.hpp:653-757) +
//                          backward/forward_extend_encoded (src/r-index.cpp:713-764) +
//                          rank_at_cached_encoded (:619-641).
//   pgx_tag_*              tag-array lookups of TagArray::query_compressed{,_compact}
//                          (src/tag_arrays.cpp:780-890): locate, gather, segmented sort-unique.
//   pgx_scan_*             device-wide exclusive scans that size / place variable-length outputs.
//
// All of it is 64-bit integer work bound by random access into the rank image (HBM / L2 / LDS);
// there is no floating point and nothing MFMA-shaped.  Wave width is hard-coded to 64.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgx_device.h"

// ------------------------------------------------------------------------------------------
// rank probe: A = count of code `cv` in BWT[0,pos), B = sum over codes of mult[code] * count(code)
// (both modulo 2^64; only differences of two probes are ever used).
template <bool LDS_IMAGE>
__device__ __forceinline__ void pgx_rank_ab(const PgxDevImage &img, const uint4 *__restrict__ lds_blocks,
                                            const uint32_t *__restrict__ lds_dir, const uint64_t *__restrict__ lds_bstart,
                                            uint64_t pos, uint32_t cv, uint32_t mrow, uint64_t &A, uint64_t &B) {
    if (pos > img.n) pos = img.n; // predecessor(pos >= size) = last block, rel past the end = totals
    const uint64_t di = pos >> img.dir_shift; // <= (n >> shift) = dir_entries - 2
    uint32_t lo, hi;
    if (LDS_IMAGE) { lo = lds_dir[di]; hi = lds_dir[di + 1]; }
    else { lo = img.dir[di]; hi = img.dir[di + 1]; }
    while (lo < hi) { // rare: only when a block boundary falls inside this directory bucket
        const uint32_t mid = (lo + hi + 1) >> 1;
        const uint64_t st = LDS_IMAGE ? lds_bstart[mid] : img.bstart[mid];
        if (st <= pos) lo = mid; else hi = mid - 1;
    }
    const uint4 *bp = (LDS_IMAGE ? lds_blocks : img.blocks) + (size_t)lo * 4;
    const uint4 h0 = bp[0], h1 = bp[1], r0 = bp[2], r1 = bp[3];
    uint64_t c[6];
    c[0] = (uint64_t)h0.x | ((uint64_t)(h1.z & 0xFFu) << 32);
    c[1] = (uint64_t)h0.y | ((uint64_t)((h1.z >> 8) & 0xFFu) << 32);
    c[2] = (uint64_t)h0.z | ((uint64_t)((h1.z >> 16) & 0xFFu) << 32);
    c[3] = (uint64_t)h0.w | ((uint64_t)(h1.z >> 24) << 32);
    c[4] = (uint64_t)h1.x | ((uint64_t)(h1.w & 0xFFu) << 32);
    c[5] = (uint64_t)h1.y | ((uint64_t)((h1.w >> 8) & 0xFFu) << 32);
    uint64_t start = 0, a = 0, b = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        start += ((img.excl_mask >> i) & 1u) ? 0ull : c[i];
        a = (cv == (uint32_t)i) ? c[i] : a;
        b += c[i] * (uint64_t)((mrow >> (3 * i)) & 7u);
    }
    uint32_t rel = (uint32_t)(pos - start); // block extent <= 16 * 8191
    uint32_t ia = 0, ib = 0;
    const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int e = 0; e < PGX_BLOCK_RUNS; e++) {
        const uint32_t v = (rw[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
        const uint32_t code = v >> PGX_RUN_LEN_BITS, len = v & PGX_RUN_LEN_MAX;
        const uint32_t take = min(len, rel);
        rel -= take;
        ia += (code == cv) ? take : 0u;
        ib += take * ((mrow >> (3 * code)) & 7u);
    }
    A = a + ia;
    B = b + ib;
}

// one FMD extension of (k, kp, s) by `byte` (backward, or forward = backward on the swapped
// interval by the complement, folded into ext_tab[256 + byte]).  Returns the new size (0 = empty).
template <bool LDS_IMAGE>
__device__ __forceinline__ void pgx_extend(const PgxDevImage &img, const uint4 *lds_blocks, const uint32_t *lds_dir,
                                           const uint64_t *lds_bstart, const uint32_t *s_ext, const uint64_t *s_C,
                                           uint64_t &k, uint64_t &kp, uint64_t &s, uint32_t byte, bool fwd) {
    const uint32_t e = s_ext[(fwd ? 256u : 0u) + byte];
    const uint32_t cv = PGX_EXT_CV(e), mrow = PGX_EXT_M(e);
    const uint64_t kk = fwd ? kp : k, kq = fwd ? k : kp;
    uint64_t A1, B1, A0, B0;
    pgx_rank_ab<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_bstart, kk + s, cv, mrow, A1, B1);
    pgx_rank_ab<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_bstart, kk, cv, mrow, A0, B0);
    if (PGX_EXT_KILL(e) || A0 >= A1) { // rank_k >= rank_ks -> bi_interval(0,0,0), src/r-index.cpp:751
        k = 0; kp = 0; s = 0;
        return;
    }
    const uint64_t nk = A0 + s_C[PGX_EXT_V(e)], nq = kq + (B1 - B0);
    s = A1 - A0;
    k = fwd ? nq : nk;
    kp = fwd ? nk : nq;
}

template <bool LDS_IMAGE>
__device__ __forceinline__ void pgx_stage_tables(const PgxDevImage &img, uint32_t *s_ext, uint64_t *s_C, uint4 *lds_blocks,
                                                 uint32_t *lds_dir, uint64_t *lds_bstart) {
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) s_ext[i] = img.consts->ext_tab[i];
    if (threadIdx.x < 8) s_C[threadIdx.x] = img.consts->C[threadIdx.x];
    if (LDS_IMAGE) {
        const uint32_t nb4 = img.n_blocks * 4;
        for (uint32_t i = threadIdx.x; i < nb4; i += blockDim.x) lds_blocks[i] = img.blocks[i];
        for (uint64_t i = threadIdx.x; i < img.dir_entries; i += blockDim.x) lds_dir[i] = img.dir[i];
        for (uint32_t i = threadIdx.x; i < img.n_blocks; i += blockDim.x) lds_bstart[i] = img.bstart[i];
    }
    __syncthreads();
}

// dynamic LDS carve (16-byte aligned base): [blocks | bstart | dir]
#define PGX_LDS_CARVE(img)                                                                   \
    extern __shared__ __align__(16) unsigned char pgx_dyn_lds[];                              \
    uint4 *lds_blocks = reinterpret_cast<uint4 *>(pgx_dyn_lds);                               \
    uint64_t *lds_bstart = reinterpret_cast<uint64_t *>(pgx_dyn_lds + (size_t)(img).n_blocks * PGX_BLOCK_BYTES); \
    uint32_t *lds_dir = reinterpret_cast<uint32_t *>(pgx_dyn_lds + (size_t)(img).n_blocks * (PGX_BLOCK_BYTES + 8))

// ------------------------------------------------------------------------------------------
// find_all_mems for a batch.  State machine of find_mems_function (algorithm.hpp:653-736):
//   phase 1  backward from j = x+min_len-1 down to x          (:666-676)
//   phase 2  forward  from j = x+min_len   up to len-1        (:684-696)  -> emit MEM (:713)
//   phase 3  backward from j = e down to x+1, fresh interval  (:718-735); pattern[len] reads 0
template <bool LDS_IMAGE>
__global__ void __launch_bounds__(PGX_FM_THREADS)
pgx_find_mems_kernel(PgxDevImage img, const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets,
                     uint64_t n_reads, uint64_t min_len, uint64_t min_occ, const uint64_t *__restrict__ slot_off,
                     pgx_mem *__restrict__ slots, uint32_t *__restrict__ mem_count, unsigned long long *__restrict__ n_ext_total) {
    __shared__ uint32_t s_ext[512];
    __shared__ uint64_t s_C[8];
    PGX_LDS_CARVE(img);
    pgx_stage_tables<LDS_IMAGE>(img, s_ext, s_C, lds_blocks, lds_dir, lds_bstart);

    const uint64_t rid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = rid < n_reads;
    uint64_t base = 0, slot = 0;
    int64_t len = 0;
    if (have) {
        base = offsets[rid];
        len = (int64_t)(offsets[rid + 1] - base);
        slot = slot_off[rid];
    }
    const uint64_t n = img.n;
    int64_t x = 0, j = 0, e = 0;
    uint64_t k = 0, kp = 0, s = 0, Jk = 0, Js = 0;
    uint32_t nm = 0, next = 0;
    int ph = 0; // 0 = done

    // begin(x): entry of find_mems_function
    auto begin = [&]() {
        if (x >= len || (uint64_t)(len - x) < min_len) { ph = 0; return; } // :745 / :658
        k = 0; kp = 0; s = n;
        if (min_len == 0) { // step 1 runs zero times (:666); step 2 starts at j = x
            Jk = 0; Js = n; j = x; ph = 2;
        } else {
            j = x + (int64_t)min_len - 1; ph = 1;
        }
    };
    // emit the MEM [x, e) and set up step 3
    auto emit = [&]() {
        pgx_mem m;
        m.start = (uint64_t)x; m.end = (uint64_t)e; m.bwt_start = Jk; m.size = (int64_t)Js;
        slots[slot + nm] = m;
        nm++;
        k = 0; kp = 0; s = n; j = e;
        if (j > x) ph = 3;
        else { x = x + 1; begin(); } // loop of :722 runs zero times, returns j + 1
    };
    if (have) begin();

    while (__any(ph != 0)) {
        if (ph != 0) {
            const uint32_t byte = (j < len) ? (uint32_t)reads[base + (uint64_t)j] : 0u;
            const bool fwd = (ph == 2);
            pgx_extend<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_bstart, s_ext, s_C, k, kp, s, byte, fwd);
            next++;
            const bool small = (s < min_occ) || (s == 0); // :671 (unsigned compare) || size <= 0
            if (ph == 1) {
                if (small) { x = j + 1; begin(); }
                else if (j == x) {
                    Jk = k; Js = s; j = x + (int64_t)min_len;
                    if (j >= len) { e = j; emit(); } else ph = 2;
                } else j--;
            } else if (ph == 2) {
                if (small) { e = j; emit(); }
                else {
                    Jk = k; Js = s; j++;
                    if (j >= len) { e = j; emit(); }
                }
            } else { // ph == 3
                if (small) { x = j + 1; begin(); }
                else {
                    j--;
                    if (j <= x) { x = x + 1; begin(); }
                }
            }
        }
    }
    if (have) mem_count[rid] = nm;
    // one atomic per wave for the extension counter
    unsigned long long tot = next;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(n_ext_total, tot);
}

template __global__ void pgx_find_mems_kernel<false>(PgxDevImage, const uint8_t *, const uint64_t *, uint64_t, uint64_t, uint64_t,
                                                     const uint64_t *, pgx_mem *, uint32_t *, unsigned long long *);
template __global__ void pgx_find_mems_kernel<true>(PgxDevImage, const uint8_t *, const uint64_t *, uint64_t, uint64_t, uint64_t,
                                                    const uint64_t *, pgx_mem *, uint32_t *, unsigned long long *);

// ------------------------------------------------------------------------------------------
// primitives for tests (mirror rank_at_cached_encoded / backward_extend_encoded / forward_...)
__global__ void __launch_bounds__(256)
pgx_rank_kernel(PgxDevImage img, const uint64_t *__restrict__ pos, uint64_t n, int true_codes, uint64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t sigma = img.consts->sigma;
    for (uint32_t sl = 0; sl < 6; sl++) {
        uint64_t A = 0, B;
        if (true_codes) pgx_rank_ab<false>(img, nullptr, nullptr, nullptr, pos[i], sl, 0, A, B);
        else if (sl < sigma) pgx_rank_ab<false>(img, nullptr, nullptr, nullptr, pos[i], img.consts->slot_code[sl], 0, A, B);
        out[i * 6 + sl] = A;
    }
}

template <bool LDS_IMAGE>
__global__ void __launch_bounds__(256)
pgx_extend_kernel(PgxDevImage img, const pgx_biint *__restrict__ in, const uint8_t *__restrict__ sym,
                  const uint8_t *__restrict__ forward, uint64_t n, pgx_biint *__restrict__ out) {
    __shared__ uint32_t s_ext[512];
    __shared__ uint64_t s_C[8];
    PGX_LDS_CARVE(img);
    pgx_stage_tables<LDS_IMAGE>(img, s_ext, s_C, lds_blocks, lds_dir, lds_bstart);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = in[i].forward, kp = in[i].reverse, s = (uint64_t)in[i].size;
    pgx_extend<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_bstart, s_ext, s_C, k, kp, s, sym[i], forward[i] != 0);
    pgx_biint o;
    o.forward = k; o.reverse = kp; o.size = (int64_t)s;
    out[i] = o;
}
template __global__ void pgx_extend_kernel<false>(PgxDevImage, const pgx_biint *, const uint8_t *, const uint8_t *, uint64_t, pgx_biint *);
template __global__ void pgx_extend_kernel<true>(PgxDevImage, const pgx_biint *, const uint8_t *, const uint8_t *, uint64_t, pgx_biint *);

// ------------------------------------------------------------------------------------------
// exclusive scan of u64 values produced by a loader (3 launches: partial sums, scan of sums, apply)
//   mode 0: in32[i]                       (u32 array)
//   mode 1: in64[i]                       (u64 array)
//   mode 2: MEM capacity of read i from offsets (min(len, len - min_len + 1), 0 if len < min_len)
__device__ __forceinline__ uint64_t pgx_scan_load(int mode, const void *in, uint64_t i, uint64_t min_len) {
    if (mode == 0) return ((const uint32_t *)in)[i];
    if (mode == 1) return ((const uint64_t *)in)[i];
    const uint64_t *off = (const uint64_t *)in;
    const uint64_t len = off[i + 1] - off[i];
    if (len < min_len) return 0;
    const uint64_t c = len - min_len + 1;
    return c < len ? c : len;
}

__device__ __forceinline__ uint64_t pgx_block_excl_scan(uint64_t v, uint64_t *s_wave, uint64_t &block_total) {
    // 256 threads = 4 waves; returns exclusive prefix of v within the block
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint64_t wbase = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < w) wbase += s_wave[i];
        tot += s_wave[i];
    }
    block_total = tot;
    __syncthreads();
    return wbase + inc - v;
}

#define PGX_SCAN_ITEMS 8 // per thread -> 2048 per block
__global__ void __launch_bounds__(256)
pgx_scan_partial_kernel(int mode, const void *in, uint64_t n, uint64_t min_len, uint64_t *__restrict__ block_sums) {
    __shared__ uint64_t s_wave[4];
    const uint64_t b0 = (uint64_t)blockIdx.x * 256 * PGX_SCAN_ITEMS;
    uint64_t v = 0;
    for (int t = 0; t < PGX_SCAN_ITEMS; t++) {
        const uint64_t i = b0 + (uint64_t)threadIdx.x * PGX_SCAN_ITEMS + t;
        if (i < n) v += pgx_scan_load(mode, in, i, min_len);
    }
    uint64_t tot;
    (void)pgx_block_excl_scan(v, s_wave, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: in-place exclusive scan of block_sums[0..nb), total appended at block_sums[nb]
__global__ void __launch_bounds__(256) pgx_scan_sums_kernel(uint64_t *block_sums, uint64_t nb) {
    __shared__ uint64_t s_wave[4];
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < nb; b0 += 256) {
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t v = i < nb ? block_sums[i] : 0;
        uint64_t tot;
        const uint64_t ex = pgx_block_excl_scan(v, s_wave, tot);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) block_sums[nb] = carry;
}

// out has n+1 entries; out[n] = total
__global__ void __launch_bounds__(256)
pgx_scan_apply_kernel(int mode, const void *in, uint64_t n, uint64_t min_len, const uint64_t *__restrict__ block_sums,
                      uint64_t nb, uint64_t *__restrict__ out) {
    __shared__ uint64_t s_wave[4];
    const uint64_t b0 = (uint64_t)blockIdx.x * 256 * PGX_SCAN_ITEMS;
    uint64_t vals[PGX_SCAN_ITEMS], v = 0;
    for (int t = 0; t < PGX_SCAN_ITEMS; t++) {
        const uint64_t i = b0 + (uint64_t)threadIdx.x * PGX_SCAN_ITEMS + t;
        vals[t] = i < n ? pgx_scan_load(mode, in, i, min_len) : 0;
        v += vals[t];
    }
    uint64_t tot;
    uint64_t ex = block_sums[blockIdx.x] + pgx_block_excl_scan(v, s_wave, tot);
    for (int t = 0; t < PGX_SCAN_ITEMS; t++) {
        const uint64_t i = b0 + (uint64_t)threadIdx.x * PGX_SCAN_ITEMS + t;
        if (i < n) out[i] = ex;
        ex += vals[t];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_sums[nb];
}

// ------------------------------------------------------------------------------------------
// MEM compaction: slots (worst-case capacity per read) -> dense CSR in read order
__global__ void __launch_bounds__(256)
pgx_compact_mems_kernel(uint64_t n_reads, const uint64_t *__restrict__ slot_off, const pgx_mem *__restrict__ slots,
                        const uint32_t *__restrict__ mem_count, const uint64_t *__restrict__ mem_off, pgx_mem *__restrict__ mems) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    const uint32_t c = mem_count[i];
    const uint64_t src = slot_off[i], dst = mem_off[i];
    for (uint32_t t = 0; t < c; t++) mems[dst + t] = slots[src + t];
}

// ------------------------------------------------------------------------------------------
// tag array.  rank_1(bwt_intervals, x + 1) = number of run starts <= x  (src/tag_arrays.cpp:857-858)
__device__ __forceinline__ uint64_t pgx_tag_rank(const PgxDevImage &img, uint64_t x) {
    const uint64_t nr = img.n_tag_runs;
    uint64_t di = x >> img.tag_dir_shift;
    if (di + 1 >= img.tag_dir_entries) return nr; // beyond bwt_intervals.size(): all ones
    uint64_t lo = img.tdir[di], hi = img.tdir[di + 1];
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (img.tstart[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// per query: run_nums (number_of_runs, :860) and the index of the first item read (:862-874,
// including the off-by-one when first_bit_index % 10 == 0)
__global__ void __launch_bounds__(256)
pgx_tag_locate_kernel(PgxDevImage img, const pgx_mem *__restrict__ mems, const uint64_t *__restrict__ qstart,
                      const uint64_t *__restrict__ qend, uint64_t n, uint64_t *__restrict__ run_nums, uint64_t *__restrict__ first_item) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t st, en;
    if (mems) { st = mems[i].bwt_start; en = st + (uint64_t)mems[i].size - 1; } // find_mems.cpp:129
    else { st = qstart[i]; en = qend[i]; }
    const uint64_t f = pgx_tag_rank(img, st), g = pgx_tag_rank(img, en);
    run_nums[i] = g - f + 1;
    first_item[i] = (f % 10) ? f - 1 : f;
}

// one wave per query: gather run values into its segment of `buf`
__global__ void __launch_bounds__(256)
pgx_tag_gather_kernel(PgxDevImage img, uint64_t n, const uint64_t *__restrict__ run_nums, const uint64_t *__restrict__ first_item,
                      const uint64_t *__restrict__ seg_off, uint64_t *__restrict__ buf, unsigned long long *__restrict__ n_overflow) {
    const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (q >= n) return;
    const uint64_t cnt = run_nums[q], src = first_item[q], dst = seg_off[q];
    bool over = false;
    for (uint64_t t = lane; t < cnt; t += 64) {
        const uint64_t it = src + t;
        uint64_t v = 0;
        if (it < img.n_tag_items) v = img.tvals[it]; else over = true; // reference reads past the end (UB): value 0
        buf[dst + t] = v;
    }
    if (__any(over) && lane == 0) atomicAdd(n_overflow, 1ull);
}

__device__ __forceinline__ void pgx_cmpswap(uint64_t &a, uint64_t &b, bool up) {
    if ((a > b) == up) { const uint64_t t = a; a = b; b = t; }
}

// bitonic sort of cnt values (padded to p2 with ~0) by ONE wave in `arr`, then duplicates dropped
// and the unique prefix written back to `seg` in chunks of 64 (write index never passes read index)
template <class Ptr>
__device__ __forceinline__ uint64_t pgx_wave_sort_unique(Ptr arr, uint64_t *__restrict__ seg, uint64_t cnt, uint64_t p2, int lane) {
    for (uint64_t t = lane; t < p2; t += 64) arr[t] = t < cnt ? seg[t] : ~0ull;
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    for (uint64_t k = 2; k <= p2; k <<= 1) {
        for (uint64_t jj = k >> 1; jj > 0; jj >>= 1) {
            for (uint64_t t = lane; t < p2 / 2; t += 64) {
                // t-th compare-exchange pair of this stage
                const uint64_t lo_i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
                const uint64_t hi_i = lo_i | jj;
                const uint64_t a = arr[lo_i], b = arr[hi_i];
                const bool up = ((lo_i & k) == 0);
                if ((a > b) == up) { arr[lo_i] = b; arr[hi_i] = a; }
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
        }
    }
    uint64_t outn = 0;
    for (uint64_t b0 = 0; b0 < cnt; b0 += 64) {
        const uint64_t t = b0 + lane;
        const uint64_t v = t < cnt ? arr[t] : 0;
        const uint64_t prev = (t > 0 && t < cnt) ? arr[t - 1] : 0;
        const bool keep = t < cnt && (t == 0 || v != prev);
        const unsigned long long mask = __ballot(keep);
        const int at = __popcll(mask & ((1ull << lane) - 1ull));
        if (keep) seg[outn + at] = v;
        outn += (uint64_t)__popcll(mask);
    }
    return outn;
}

// one wave per query: sort its segment and drop duplicates in place; ucount[q] = #unique.
//   cnt <= 64        bitonic network in registers (cross-lane shuffles)
//   cnt <= LDS_CAP   bitonic in LDS (per-wave slice)
//   otherwise        bitonic in global memory (rare: a short MEM with a huge SA interval)
#define PGX_SORT_LDS_CAP 2048
__global__ void __launch_bounds__(256)
pgx_tag_sort_unique_kernel(uint64_t n, const uint64_t *__restrict__ run_nums, const uint64_t *__restrict__ seg_off,
                           uint64_t *__restrict__ buf, uint64_t *__restrict__ scratch, const uint64_t *__restrict__ scratch_off,
                           uint64_t *__restrict__ ucount) {
    __shared__ uint64_t s_sort[4][PGX_SORT_LDS_CAP];
    const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (q >= n) return;
    const uint64_t cnt = run_nums[q];
    uint64_t *seg = buf + seg_off[q];
    if (cnt <= 64) {
        uint64_t v = (uint64_t)lane < cnt ? seg[lane] : ~0ull;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
                const uint64_t o = __shfl_xor(v, jj, 64);
                const bool up = ((lane & k) == 0);
                const bool lower = ((lane & jj) == 0);
                const uint64_t mn = v < o ? v : o, mx = v < o ? o : v;
                v = (lower == up) ? mn : mx;
            }
        }
        const uint64_t prev = __shfl_up(v, 1, 64);
        const bool keep = (uint64_t)lane < cnt && (lane == 0 || v != prev);
        const unsigned long long mask = __ballot(keep);
        const int at = __popcll(mask & ((1ull << lane) - 1ull));
        if (keep) seg[at] = v;
        if (lane == 0) ucount[q] = (uint64_t)__popcll(mask);
        return;
    }
    // power-of-two padded bitonic sort by one wave (LDS slice or global scratch); the two calls are
    // separate inlined copies so that each keeps a statically known address space
    uint64_t p2 = 64;
    while (p2 < cnt) p2 <<= 1;
    uint64_t outn;
    if (p2 <= PGX_SORT_LDS_CAP) outn = pgx_wave_sort_unique(&s_sort[w][0], seg, cnt, p2, lane);
    else outn = pgx_wave_sort_unique(scratch + scratch_off[q], seg, cnt, p2, lane);
    if (lane == 0) ucount[q] = outn;
}

// one wave per query: copy the unique prefix of its segment to the dense positions array
__global__ void __launch_bounds__(256)
pgx_tag_compact_kernel(uint64_t n, const uint64_t *__restrict__ ucount, const uint64_t *__restrict__ seg_off,
                       const uint64_t *__restrict__ buf, const uint64_t *__restrict__ pos_off, uint64_t *__restrict__ positions) {
    const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (q >= n) return;
    const uint64_t c = ucount[q], src = seg_off[q], dst = pos_off[q];
    for (uint64_t t = lane; t < c; t += 64) positions[dst + t] = buf[src + t];
}

// scratch requirement of the global bitonic path: next pow2 of cnt when it exceeds the LDS cap
__global__ void __launch_bounds__(256)
pgx_tag_scratch_need_kernel(uint64_t n, const uint64_t *__restrict__ run_nums, uint64_t *__restrict__ need) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t cnt = run_nums[i];
    uint64_t p2 = 64;
    while (p2 < cnt) p2 <<= 1;
    need[i] = p2 > PGX_SORT_LDS_CAP ? p2 : 0;
}

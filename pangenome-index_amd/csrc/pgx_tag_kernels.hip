// pgx_tag_kernels.hip -- tag-array lookups of find_mems (TagArray::query_compressed{,_compact},
// src/tag_arrays.cpp:780-890, minus the printing) as gfx950 kernels.
//
//   locate     per MEM: two rank_1 on bwt_intervals (= searches in the sorted run-start array via a
//              sampled directory) -> run_nums, first item; MEMs with more than 16 runs go on a list
//   small      <= 16 runs (the common case: a MEM's SA interval spans ~#haplotypes positions):
//              16 lanes per MEM, four MEMs per wavefront; gather + bitonic network in registers +
//              unique, no LDS, full occupancy
//   big        listed MEMs with 17..2048 runs, one wavefront each: gather, then bitonic in registers
//              (<= 64) or in a per-wave LDS slice
//   large      listed MEMs with more runs (a MEM inside a repeat / an N run has a huge SA interval):
//              one 1024-thread workgroup each, bitonic in 128 KiB of LDS (<= 16384 values) or in
//              global scratch beyond that
//   compact    16 lanes per MEM copy the unique prefix to the dense positions array
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgx_device.h"

// run start r: from the interleaved (start, value) array when the image has one -- the locate kernel then finds the run's value in the
// cache line its search ended in (three random lines per MEM -> two)
#define PGX_TSTART(img, r) ((img).tpair ? (img).tpair[(r)].x : (img).tstart[(r)])
// rank_1(bwt_intervals, x + 1) = number of run starts <= x  (src/tag_arrays.cpp:857-858)
__device__ __forceinline__ uint64_t pgx_tag_rank(const PgxDevImage &img, uint64_t x) {
    const uint64_t nr = img.n_tag_runs;
    uint64_t di = x >> img.tag_dir_shift;
    if (di + 1 >= img.tag_dir_entries) return nr; // beyond bwt_intervals.size(): all ones
    uint64_t lo = img.tdir[di], hi = img.tdir[di + 1];
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (PGX_TSTART(img, mid) <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Tag runs by bucket (round 3): the locate kernel is two to three dependent random lines per MEM through tdir and tpair (directory entry, the pair the
// search ends at, sometimes a neighbour) -- 1.24 ms for 18.8 M MEMs at chr22 scale, at the rate random lines reach.  A bucket is the 2^shift BWT positions
// [b << shift, (b + 1) << shift) and ONE 128-byte line:
//   dw 0        R0 = number of run starts below the bucket
//   dw 1        c = run starts inside it (bits 0..7; 255: more than PGX_TBUCKET_RUNS, the line holds nothing else)
//   dw 2..6     the c starts as 16-bit offsets into the bucket, ascending
//   dw 8..31    the values of items R0 - 1 .. R0 + c (u64 each; 0 where there is no such item): whichever item a one-run query reads (f - 1, or f
//               itself where f % 10 == 0) is in the line
// so a MEM whose interval stays inside one bucket (nearly all: shift is chosen for ~4 runs per bucket) costs one line.
__global__ void __launch_bounds__(256)
pgx_tag_bucket_kernel(const uint64_t *__restrict__ tstart, const uint64_t *__restrict__ tvals, uint64_t n_runs, uint64_t n_items, uint32_t shift,
                      uint64_t n_buckets, uint4 *__restrict__ out) {
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p0 = b << shift, p1 = (b + 1) << shift;
        uint64_t lo = 0, hi = n_runs;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (tstart[mid] < p0) lo = mid + 1; else hi = mid; }
        const uint64_t r0 = lo;
        hi = n_runs;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (tstart[mid] < p1) lo = mid + 1; else hi = mid; }
        const uint64_t c = lo - r0;
        uint32_t dw[32];
#pragma unroll
        for (int i = 0; i < 32; i++) dw[i] = 0u;
        dw[0] = (uint32_t)r0;
        if (c > PGX_TBUCKET_RUNS) dw[1] = 255u;
        else {
            dw[1] = (uint32_t)c;
#pragma unroll
            for (uint32_t i = 0; i < PGX_TBUCKET_RUNS; i++)
                if (i < c) dw[2 + (i >> 1)] |= (uint32_t)(tstart[r0 + i] - p0) << (16u * (i & 1u));
#pragma unroll
            for (uint32_t i = 0; i < PGX_TBUCKET_RUNS + 2u; i++) {
                const uint64_t it = r0 + i; // item it - 1
                const uint64_t v = (i <= c + 1 && it >= 1 && it - 1 < n_items) ? tvals[it - 1] : 0ull;
                dw[8 + 2 * i] = (uint32_t)v; dw[9 + 2 * i] = (uint32_t)(v >> 32);
            }
        }
        uint4 *o = out + b * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = make_uint4(dw[4 * i], dw[4 * i + 1], dw[4 * i + 2], dw[4 * i + 3]);
    }
}
// number of the bucket's run starts <= off (the 16-bit offsets of dw 2..6)
__device__ __forceinline__ uint32_t pgx_tbucket_le(const uint4 &h0, const uint4 &h1, uint32_t c, uint32_t off) {
    const uint32_t w[5] = {h0.z, h0.w, h1.x, h1.y, h1.z};
    uint32_t k = 0;
#pragma unroll
    for (uint32_t i = 0; i < PGX_TBUCKET_RUNS; i++) {
        const uint32_t s = (w[i >> 1] >> (16u * (i & 1u))) & 0xFFFFu;
        k += (i < c && s <= off) ? 1u : 0u;
    }
    return k;
}

// per query: run_nums (number_of_runs, :860) and the index of the first item read (:862-874,
// including the off-by-one when first_bit_index % 10 == 0)
// Counts that live on the device: every kernel of the tag stage takes its element count as a value (an upper bound: the
// capacity its buffers were sized for) AND, optionally, a device pointer to the actual count, loops over its elements with a
// grid stride, and returns at once when the stage's abort flag is set -- so the whole stage can be enqueued without the host
// knowing any count (pgx_runtime.hip "speculative sizing"); with a NULL pointer and an exact value it is the plain launch.
#define PGX_DEV_COUNT(n, n_dev) ((n_dev) ? ((*(n_dev)) < (n) ? (*(n_dev)) : (n)) : (n))
#define PGX_ABORTED(abort) ((abort) && *(abort))

__global__ void __launch_bounds__(PGX_TAG_LOCATE_THREADS)
pgx_tag_locate_kernel(PgxDevImage img, const pgx_mem *__restrict__ mems, const uint64_t *__restrict__ qstart,
                      const uint64_t *__restrict__ qend, uint64_t n_cap, const uint64_t *__restrict__ n_dev, const uint64_t *__restrict__ abort,
                      uint64_t *__restrict__ run_nums,
                      uint64_t *__restrict__ first_item, uint64_t *__restrict__ need, uint64_t *__restrict__ big_list, uint64_t *__restrict__ large_list,
                      unsigned long long *__restrict__ n_big, unsigned long long *__restrict__ n_large, uint64_t *__restrict__ single,
                      uint64_t *__restrict__ ucount, unsigned long long *__restrict__ n_overflow, uint64_t *__restrict__ small_list,
                      unsigned long long *__restrict__ n_small) {
  if (PGX_ABORTED(abort)) return;
  const uint64_t n = PGX_DEV_COUNT(n_cap, n_dev);
  __shared__ uint32_t s_cnt[PGX_TAG_LOCATE_THREADS / 64];
  __shared__ unsigned long long s_base;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += (uint64_t)gridDim.x * blockDim.x) { // (uniform per workgroup)
    const uint64_t i = base + threadIdx.x;
    const bool valid = i < n; // no early return: the small list is appended to by whole waves
    uint64_t cnt = 0;
    if (valid) {
        uint64_t st, en;
        if (mems) { st = mems[i].bwt_start; en = st + (uint64_t)mems[i].size - 1; } // find_mems.cpp:129
        else { st = qstart[i]; en = qend[i]; }
        // through the bucket of `st` where there is one (and it is not flagged): f, usually g, and a one-run query's value from a single line
        bool done = false;
        uint64_t f = 0, g = 0, bval = 0;
        bool bval_ok = false;
        if (img.tbucket && en >= st && (st >> img.tbucket_shift) < img.n_tbuckets) {
            const uint32_t sh = img.tbucket_shift, msk = (1u << sh) - 1u;
            const uint64_t b0 = st >> sh, b1 = en >> sh;
            const uint4 *bp = img.tbucket + b0 * 8;
            const uint4 h0 = bp[0], h1 = bp[1];
            const uint32_t c0 = h0.y & 0xFFu;
            if (c0 != 255u) {
                f = (uint64_t)h0.x + pgx_tbucket_le(h0, h1, c0, (uint32_t)st & msk);
                if (b1 == b0) { g = (uint64_t)h0.x + pgx_tbucket_le(h0, h1, c0, (uint32_t)en & msk); done = true; }
                else if (b1 < img.n_tbuckets) {
                    const uint4 *ep = img.tbucket + b1 * 8;
                    const uint4 e0 = ep[0], e1 = ep[1];
                    const uint32_t c1 = e0.y & 0xFFu;
                    if (c1 != 255u) { g = (uint64_t)e0.x + pgx_tbucket_le(e0, e1, c1, (uint32_t)en & msk); done = true; }
                }
                if (done && g == f) { // one run: its item's value is in the line of b0 (items R0 - 1 .. R0 + c)
                    const uint64_t fi0 = (f % 10) ? f - 1 : f;
                    const uint32_t slot = (uint32_t)(fi0 + 1 - (uint64_t)h0.x); // 0 .. c + 1
                    const uint2 *vp = reinterpret_cast<const uint2 *>(bp) + 4 + slot;
                    const uint2 vv = *vp;
                    bval = (uint64_t)vv.x | ((uint64_t)vv.y << 32);
                    bval_ok = true;
                }
            }
        }
        if (!done) {
        f = pgx_tag_rank(img, st);
        // rank of `en`: an interval usually ends inside the run it starts in, so gallop upwards from f (one load when it does)
        g = f;
        if (en >= st) {
            const uint64_t nr = img.n_tag_runs;
            uint64_t step = 1, lo = f, hi = f;
            while (hi < nr && PGX_TSTART(img, hi) <= en) { lo = hi + 1; hi = (hi + step < nr) ? hi + step : nr; step <<= 1; }
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (PGX_TSTART(img, mid) <= en) lo = mid + 1; else hi = mid;
            }
            g = lo;
        } else g = pgx_tag_rank(img, en);
        }
        cnt = g - f + 1;
        run_nums[i] = cnt;
        const uint64_t fi = (f % 10) ? f - 1 : f;
        first_item[i] = fi;
        if (cnt == 1) { // one run = one position: the common case (a MEM inside one node); no segment, no sort
            uint64_t v = 0;
            if (fi < img.n_tag_items) v = bval_ok ? bval : (img.tpair ? img.tpair[fi].y : img.tvals[fi]);
            else atomicAdd(n_overflow, 1ull); // the reference reads past the stored runs (UB there): value 0
            single[i] = v;
            ucount[i] = 1;
        } else if (cnt == 0) ucount[i] = 0; // an inverted query is on no list
        uint64_t p2 = 64;
        while (p2 < cnt && p2 < (1ull << 62)) p2 <<= 1;
        need[i] = p2 > PGX_SORT_WG_LDS_CAP ? p2 : 0; // global scratch of the large path
        if (cnt > PGX_SORT_LDS_CAP) {
            large_list[atomicAdd(n_large, 1ull)] = i;
            atomicMax(n_large + 1, (unsigned long long)cnt); // sizes the LDS of the large-path launch
        }
        else if (cnt > PGX_TAG_SMALL) big_list[atomicAdd(n_big, 1ull)] = i;
    }
    // queries with 2 .. 16 runs: a list of their own, so that the 16-lane kernels only see those.  One atomic per
    // 1024-thread workgroup: all of them hit one address, and one per wave (23 k for 1.5 M MEMs) cost 0.27 ms.
    const bool is_small = valid && cnt >= 2 && cnt <= PGX_TAG_SMALL;
    const unsigned long long m = __ballot(is_small);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < PGX_TAG_LOCATE_THREADS / 64; k++) { const uint32_t c = s_cnt[k]; s_cnt[k] = tot; tot += c; }
        s_base = tot ? atomicAdd(n_small, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    if (is_small) small_list[s_base + s_cnt[wv] + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull))] = i;
    __syncthreads(); // s_cnt / s_base are reused by the next round
  }
}

// 16 lanes per query, <= 16 runs: gather, sort, unique -> seg, ucount
__global__ void __launch_bounds__(256)
pgx_tag_small_kernel(PgxDevImage img, const uint64_t *__restrict__ list, uint64_t n_cap, const uint64_t *__restrict__ n_dev,
                     const uint64_t *__restrict__ abort, const uint64_t *__restrict__ run_nums,
                     const uint64_t *__restrict__ first_item, const uint64_t *__restrict__ seg_off, uint64_t *__restrict__ buf,
                     uint64_t *__restrict__ ucount, unsigned long long *__restrict__ n_overflow) {
  if (PGX_ABORTED(abort)) return;
  const uint64_t n = PGX_DEV_COUNT(n_cap, n_dev);
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4;
  for (uint64_t wbase = ((uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) >> 4; wbase < n; wbase += ((uint64_t)gridDim.x * blockDim.x) >> 4) { // (uniform per wave)
    const uint64_t w16 = wbase + (uint64_t)grp; // entry of the small list (2 .. 16 runs)
    const bool valid = w16 < n;
    const uint64_t q = valid ? list[w16] : 0;
    const uint64_t cnt = valid ? run_nums[q] : 0;
    const bool mine = valid;
    uint64_t v = ~0ull;
    bool over = false;
    if (mine && (uint64_t)l16 < cnt) {
        const uint64_t it = first_item[q] + (uint64_t)l16;
        if (it < img.n_tag_items) v = img.tvals[it];
        else { v = 0; over = true; } // the reference reads past the stored runs (UB): value 0
    }
#pragma unroll
    for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            const uint64_t o = __shfl_xor(v, jj, 64);
            const bool up = ((l16 & k) == 0), lower = ((l16 & jj) == 0);
            const uint64_t mn = v < o ? v : o, mx = v < o ? o : v;
            v = (lower == up) ? mn : mx;
        }
    }
    const uint64_t prev = __shfl_up(v, 1, 64);
    const bool keep = mine && (uint64_t)l16 < cnt && (l16 == 0 || v != prev);
    const unsigned long long mask = __ballot(keep);
    const unsigned long long omask = __ballot(over);
    const uint32_t gm = (uint32_t)(mask >> (16 * grp)) & 0xFFFFu;
    if (keep) buf[seg_off[q] + (uint64_t)__popc(gm & ((1u << l16) - 1u))] = v;
    if (mine && l16 == 0) {
        ucount[q] = (uint64_t)__popc(gm);
        if ((omask >> (16 * grp)) & 0xFFFFull) atomicAdd(n_overflow, 1ull);
    }
  }
}

// one wave per listed query: gather run values into its segment of `buf`
__global__ void __launch_bounds__(256)
pgx_tag_gather_kernel(PgxDevImage img, const uint64_t *__restrict__ list, uint64_t n_cap, const uint64_t *__restrict__ n_dev,
                      const uint64_t *__restrict__ abort, const uint64_t *__restrict__ run_nums,
                      const uint64_t *__restrict__ first_item, const uint64_t *__restrict__ seg_off, uint64_t *__restrict__ buf,
                      unsigned long long *__restrict__ n_overflow) {
  if (PGX_ABORTED(abort)) return;
  const uint64_t n_list = PGX_DEV_COUNT(n_cap, n_dev);
  const int lane = threadIdx.x & 63;
  for (uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_list; w += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
    const uint64_t q = list[w];
    const uint64_t cnt = run_nums[q], src = first_item[q], dst = seg_off[q];
    bool over = false;
    for (uint64_t t = lane; t < cnt; t += 64) {
        const uint64_t it = src + t;
        uint64_t v = 0;
        if (it < img.n_tag_items) v = img.tvals[it]; else over = true;
        buf[dst + t] = v;
    }
    if (__any(over) && lane == 0) atomicAdd(n_overflow, 1ull);
  }
}

// bitonic sort of cnt values (padded to p2 with ~0) by ONE wave in `arr`, then duplicates dropped
// and the unique prefix written back to `seg` in chunks of 64 (write index never passes read index).
// Instantiated once per address space (LDS slice / global scratch): generic pointers that may
// point into LDS are avoided on purpose (flat LDS accesses faulted on the gfx950 boxes).
template <class Ptr>
__device__ __forceinline__ uint64_t pgx_wave_sort_unique(Ptr arr, uint64_t *__restrict__ seg, uint64_t cnt, uint64_t p2, int lane) {
    for (uint64_t t = lane; t < p2; t += 64) arr[t] = t < cnt ? seg[t] : ~0ull;
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    for (uint64_t k = 2; k <= p2; k <<= 1) {
        for (uint64_t jj = k >> 1; jj > 0; jj >>= 1) {
            for (uint64_t t = lane; t < p2 / 2; t += 64) {
                const uint64_t lo_i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)); // t-th pair of this stage
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

// one wave per listed query: sort its segment and drop duplicates in place; ucount[q] = #unique
__global__ void __launch_bounds__(256)
pgx_tag_sort_unique_kernel(const uint64_t *__restrict__ list, uint64_t n_cap, const uint64_t *__restrict__ n_dev, const uint64_t *__restrict__ abort,
                           const uint64_t *__restrict__ run_nums,
                           const uint64_t *__restrict__ seg_off, uint64_t *__restrict__ buf, uint64_t *__restrict__ ucount) {
  __shared__ uint64_t s_sort[4][PGX_SORT_LDS_CAP];
  if (PGX_ABORTED(abort)) return;
  const uint64_t n_list = PGX_DEV_COUNT(n_cap, n_dev);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint64_t wi = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wi < n_list; wi += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
    const uint64_t q = list[wi];
    const uint64_t cnt = run_nums[q];
    uint64_t *seg = buf + seg_off[q];
    if (cnt <= 64) {
        uint64_t v = (uint64_t)lane < cnt ? seg[lane] : ~0ull;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
                const uint64_t o = __shfl_xor(v, jj, 64);
                const bool up = ((lane & k) == 0), lower = ((lane & jj) == 0);
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
        continue;
    }
    uint64_t p2 = 64;
    while (p2 < cnt) p2 <<= 1;
    const uint64_t outn = pgx_wave_sort_unique(&s_sort[w][0], seg, cnt, p2, lane); // cnt <= PGX_SORT_LDS_CAP by construction
    if (lane == 0) ucount[q] = outn;
  }
}

// bitonic sort of cnt values (padded to p2 with ~0) by ONE workgroup in `arr` (LDS or global
// scratch; one instantiation per address space), duplicates dropped, unique prefix written to `seg`.
template <class Ptr>
__device__ __forceinline__ uint64_t pgx_block_sort_unique(Ptr arr, uint64_t *__restrict__ seg, uint64_t cnt, uint64_t p2,
                                                          uint32_t *s_wsum) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    for (uint64_t t = tid; t < p2; t += nt) arr[t] = t < cnt ? seg[t] : ~0ull;
    __syncthreads();
    for (uint64_t k = 2; k <= p2; k <<= 1) {
        for (uint64_t jj = k >> 1; jj > 0; jj >>= 1) {
            for (uint64_t t = tid; t < p2 / 2; t += nt) {
                const uint64_t lo_i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
                const uint64_t hi_i = lo_i | jj;
                const uint64_t a = arr[lo_i], b = arr[hi_i];
                const bool up = ((lo_i & k) == 0);
                if ((a > b) == up) { arr[lo_i] = b; arr[hi_i] = a; }
            }
            __syncthreads();
        }
    }
    uint64_t outn = 0;
    for (uint64_t b0 = 0; b0 < cnt; b0 += nt) {
        const uint64_t t = b0 + tid;
        const uint64_t v = t < cnt ? arr[t] : 0;
        const uint64_t prev = (t > 0 && t < cnt) ? arr[t - 1] : 0;
        const bool keep = t < cnt && (t == 0 || v != prev);
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) s_wsum[wv] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t i = 0; i < nw; i++) {
            const uint32_t c = s_wsum[i];
            before += i < wv ? c : 0u;
            total += c;
        }
        if (keep) seg[outn + before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = v;
        outn += total;
        __syncthreads();
    }
    return outn;
}

// one 1024-thread workgroup per listed query with more than PGX_SORT_LDS_CAP runs
__global__ void __launch_bounds__(1024)
pgx_tag_sort_large_kernel(const uint64_t *__restrict__ list, uint64_t n_cap, const uint64_t *__restrict__ n_dev, const uint64_t *__restrict__ abort,
                          const uint64_t *__restrict__ run_nums,
                          const uint64_t *__restrict__ seg_off, uint64_t *__restrict__ buf, uint64_t *__restrict__ scratch,
                          const uint64_t *__restrict__ scratch_off, uint64_t *__restrict__ ucount) {
    extern __shared__ __align__(16) uint64_t pgx_sort_lds[]; // PGX_SORT_WG_LDS_CAP values
    __shared__ uint32_t s_wsum[16];
    if (PGX_ABORTED(abort)) return;
    const uint64_t n_list = PGX_DEV_COUNT(n_cap, n_dev);
    for (uint64_t e = blockIdx.x; e < n_list; e += gridDim.x) {
        const uint64_t q = list[e];
        const uint64_t cnt = run_nums[q];
        uint64_t *seg = buf + seg_off[q];
        uint64_t p2 = 64;
        while (p2 < cnt) p2 <<= 1;
        uint64_t outn;
        if (p2 <= PGX_SORT_WG_LDS_CAP) outn = pgx_block_sort_unique(&pgx_sort_lds[0], seg, cnt, p2, s_wsum);
        else outn = pgx_block_sort_unique(scratch + scratch_off[q], seg, cnt, p2, s_wsum);
        if (threadIdx.x == 0) ucount[q] = outn;
        __syncthreads();
    }
}

// 16 lanes per query: copy the unique prefix of its segment to the dense positions array.  Segments with more than
// `max_count` values are left to pgx_tag_compact_list_kernel (a few huge segments would otherwise keep 16 lanes busy
// for thousands of iterations while the rest of the grid has finished).
__global__ void __launch_bounds__(256)
pgx_tag_compact_kernel(const uint64_t *__restrict__ list, uint64_t n_cap, const uint64_t *__restrict__ n_dev, const uint64_t *__restrict__ abort,
                       const uint64_t *__restrict__ ucount, const uint64_t *__restrict__ seg_off,
                       const uint64_t *__restrict__ buf, const uint64_t *__restrict__ pos_off, uint64_t *__restrict__ positions,
                       uint64_t max_count) {
    if (PGX_ABORTED(abort)) return;
    const uint64_t n = PGX_DEV_COUNT(n_cap, n_dev);
    const int l16 = threadIdx.x & 15;
    for (uint64_t w16 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; w16 < n; w16 += ((uint64_t)gridDim.x * blockDim.x) >> 4) {
        const uint64_t q = list ? list[w16] : w16; // a list of queries (small / big / large), or all of them
        const uint64_t c = ucount[q], src = seg_off[q], dst = pos_off[q];
        if (c > max_count) continue;
        for (uint64_t t = l16; t < c; t += 16) positions[dst + t] = buf[src + t];
    }
}

// one thread per query: single-run queries (answered by the locate kernel) go straight to their place
__global__ void __launch_bounds__(256)
pgx_tag_compact_single_kernel(uint64_t n_cap, const uint64_t *__restrict__ n_dev, const uint64_t *__restrict__ abort, const uint64_t *__restrict__ run_nums,
                              const uint64_t *__restrict__ single, const uint64_t *__restrict__ pos_off, uint64_t *__restrict__ positions) {
    if (PGX_ABORTED(abort)) return;
    const uint64_t n = PGX_DEV_COUNT(n_cap, n_dev);
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x)
        if (run_nums[q] == 1) positions[pos_off[q]] = single[q];
}

// one workgroup per listed query: the segments pgx_tag_compact_kernel skipped (more than `max_count` values)
__global__ void __launch_bounds__(256)
pgx_tag_compact_list_kernel(const uint64_t *__restrict__ list, uint64_t n_cap, const uint64_t *__restrict__ n_dev, const uint64_t *__restrict__ abort,
                            const uint64_t *__restrict__ ucount,
                            const uint64_t *__restrict__ seg_off, const uint64_t *__restrict__ buf, const uint64_t *__restrict__ pos_off,
                            uint64_t *__restrict__ positions, uint64_t max_count) {
    if (PGX_ABORTED(abort)) return;
    const uint64_t n_list = PGX_DEV_COUNT(n_cap, n_dev);
    for (uint64_t e = blockIdx.x; e < n_list; e += gridDim.x) {
        const uint64_t q = list[e];
        const uint64_t c = ucount[q], src = seg_off[q], dst = pos_off[q];
        if (c <= max_count) continue;
        for (uint64_t t = threadIdx.x; t < c; t += blockDim.x) positions[dst + t] = buf[src + t];
    }
}

// Identical large queries are common (every read from the same repeat / N run yields the same SA interval): one representative
// per distinct (first item, run count) is sorted, the others copy its result.  Grouping on the device: open addressing over
// `table` (zeroed by the host, a power of two of slots >= 2 x the list), slot = 1 + the query id of the first query that claimed it;
// whichever duplicate wins the slot becomes the representative -- their results are identical, so the output does not depend on it.
__global__ void __launch_bounds__(256)
pgx_tag_dedup_kernel(const uint64_t *__restrict__ list, uint64_t n_cap, const uint64_t *__restrict__ n_dev, const uint64_t *__restrict__ abort,
                     const uint64_t *__restrict__ first_item, const uint64_t *__restrict__ run_nums, unsigned long long *__restrict__ table,
                     uint64_t table_mask, uint64_t *__restrict__ reps, unsigned long long *__restrict__ n_rep, uint64_t *__restrict__ pairs,
                     unsigned long long *__restrict__ n_dup) {
    if (PGX_ABORTED(abort)) return;
    const uint64_t n_list = PGX_DEV_COUNT(n_cap, n_dev);
    // (whole waves walk the list, so that the two global counters take ONE atomic per wave and round instead of one per query: with many identical
    //  large queries -- every read cut from the same N run asks the same one -- half a million single atomics on two words were 1.1 ms of a step)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t e0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); e0 < n_list; e0 += stride) {
        const uint64_t e = e0 + lane;
        const bool valid = e < n_list;
        uint64_t q = 0, r = 0;
        int kind = 0; // 1: representative, 2: duplicate of r
        if (valid) {
            q = list[e];
            const uint64_t f = first_item[q], c = run_nums[q];
            uint64_t h = (f * 0x9E3779B97F4A7C15ull ^ c * 0xC2B2AE3D27D4EB4Full) & table_mask;
            for (;; h = (h + 1) & table_mask) {
                unsigned long long prev = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (a slot never changes once it is claimed)
                if (prev == 0ull) prev = atomicCAS(&table[h], 0ull, (unsigned long long)(q + 1));
                if (prev == 0ull) { kind = 1; break; }
                r = (uint64_t)prev - 1;
                if (first_item[r] == f && run_nums[r] == c) { kind = 2; break; }
            }
        }
        const unsigned long long m_rep = __ballot(kind == 1), m_dup = __ballot(kind == 2);
        unsigned long long b_rep = 0, b_dup = 0;
        if (lane == 0) {
            if (m_rep) b_rep = atomicAdd(n_rep, (unsigned long long)__popcll(m_rep));
            if (m_dup) b_dup = atomicAdd(n_dup, (unsigned long long)__popcll(m_dup));
        }
        b_rep = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b_rep >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b_rep);
        b_dup = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b_dup >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b_dup);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (kind == 1) reps[b_rep + (unsigned long long)__popcll(m_rep & below)] = q;
        if (kind == 2) {
            const unsigned long long at = b_dup + (unsigned long long)__popcll(m_dup & below);
            pairs[2 * at] = q; pairs[2 * at + 1] = r;
        }
    }
}

// a (duplicate, representative) pair: the duplicate query reads exactly the same items, so its sorted unique result is the representative's (and it
// overflows iff the representative does).  Nothing is copied here (round 4): the duplicate's segment offset is pointed at the representative's
// values, and the compaction, which only ever reads seg_off[q] as the start of q's values, copies them from there straight to their final place
// (before, every duplicate's list was written twice: 0.7 ms of a step with 5 % of the reads cut from N runs).
__global__ void __launch_bounds__(256)
pgx_tag_copy_dups_kernel(const uint64_t *__restrict__ pairs, uint64_t n_cap, const uint64_t *__restrict__ n_dev, const uint64_t *__restrict__ abort,
                         uint64_t n_tag_items,
                         const uint64_t *__restrict__ first_item, const uint64_t *__restrict__ run_nums,
                         uint64_t *__restrict__ seg_off, uint64_t *__restrict__ buf, uint64_t *__restrict__ ucount,
                         unsigned long long *__restrict__ n_overflow) {
    if (PGX_ABORTED(abort)) return;
    const uint64_t n_pairs = PGX_DEV_COUNT(n_cap, n_dev);
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_pairs; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t dup = pairs[2 * e], rep = pairs[2 * e + 1];
        ucount[dup] = ucount[rep];
        seg_off[dup] = seg_off[rep]; // (a representative is never a duplicate: its own entry stays)
        if (first_item[dup] + run_nums[dup] > n_tag_items) atomicAdd(n_overflow, 1ull);
    }
    (void)buf;
}

// capacity check of a speculatively sized stage: raises the abort flag when a count the following kernels rely on exceeds what
// their buffers were sized for (the host then repeats the run with exact sizes)
__global__ void pgx_spec_check_kernel(const uint64_t *__restrict__ v0, uint64_t c0, const uint64_t *__restrict__ v1, uint64_t c1,
                                      const uint64_t *__restrict__ v2, uint64_t c2, const uint64_t *__restrict__ v3, uint64_t c3, uint64_t *__restrict__ abort) {
    if (threadIdx.x) return;
    unsigned long long bits = 0;
    if (v0 && *v0 > c0) bits |= 1ull;
    if (v1 && *v1 > c1) bits |= 2ull;
    if (v2 && *v2 > c2) bits |= 4ull;
    if (v3 && *v3 > c3) bits |= 8ull;
    if (bits) atomicOr((unsigned long long *)abort, bits);
}

// (start, value) of every tag run side by side (see PGX_TSTART)
__global__ void __launch_bounds__(256)
pgx_tag_pair_kernel(const uint64_t *__restrict__ tstart, const uint64_t *__restrict__ tvals, uint64_t n_runs, uint64_t n_items, ulonglong2 *__restrict__ out) {
    const uint64_t n = n_runs > n_items ? n_runs : n_items;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        ulonglong2 v;
        v.x = i < n_runs ? tstart[i] : ~0ull;
        v.y = i < n_items ? tvals[i] : 0ull;
        out[i] = v;
    }
}

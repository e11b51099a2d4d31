// pgx_locate_kernels.hip -- the r-index locate path as gfx950 kernels:
// FastLocate::locateNext (src/r-index.cpp:1363-1366), locate / locate_encoded (:1252-1341) and
// decompressSA / decompressDA (:1343-1361).
//
// The reference walks one chain per query: first = samples[run of state.first], then one locateNext per BWT
// position.  Every BWT run starts with a stored sample, so the chain is cut at run boundaries here and each
// (query, run) piece is walked by its own lane -- decompressSA becomes r independent chains instead of one chain of
// n steps.  A step is a predecessor search in the sorted tail positions (directory load + a short binary search)
// and one load of the next head sample: dependent random 8-byte reads, latency-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgx_device.h"

// number of elements <= x in the ascending array `arr` (cnt entries) with sampled directory `dir`
__device__ __forceinline__ uint64_t pgx_sorted_upper(const uint64_t *__restrict__ arr, const uint32_t *__restrict__ dir,
                                                     uint32_t shift, uint64_t entries, uint64_t cnt, uint64_t x) {
    const uint64_t di = x >> shift;
    if (di + 1 >= entries) return cnt;
    uint64_t lo = dir[di], hi = dir[di + 1];
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (arr[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// FastLocate::locateNext: samples[last_to_run[pred] + 1] + (prev - pred position)
__device__ __forceinline__ uint64_t pgx_locate_next(const PgxLocImage &loc, uint64_t prev) {
    if (prev == PGX_NO_POSITION) return PGX_NO_POSITION;
    const uint64_t c = pgx_sorted_upper(loc.lpos, loc.ldir, loc.ldir_shift, loc.ldir_entries, loc.n_last, prev);
    if (c == 0) return PGX_NO_POSITION; // no tail sample at or before prev: undefined in the reference
    const uint64_t nx = loc.lnext[c - 1];
    return nx == PGX_NO_POSITION ? PGX_NO_POSITION : nx + (prev - loc.lpos[c - 1]);
}

__global__ void __launch_bounds__(256)
pgx_locate_next_kernel(PgxLocImage loc, const uint64_t *__restrict__ prev, uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pgx_locate_next(loc, prev[i]);
}

// per query [qs, qe]: the run holding qs and the number of runs the range touches (0 for an empty range)
__global__ void __launch_bounds__(256)
pgx_locate_plan_kernel(PgxLocImage loc, const uint64_t *__restrict__ qs, const uint64_t *__restrict__ qe, uint64_t n,
                       uint64_t *__restrict__ run0, uint64_t *__restrict__ n_pieces) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = qs[i], b = qe[i];
    if (b < a) { run0[i] = 0; n_pieces[i] = 0; return; } // state.second < state.first -> empty result (:1255)
    const uint64_t r0 = pgx_sorted_upper(loc.rstart, loc.rdir, loc.rdir_shift, loc.rdir_entries, loc.n_runs, a) - 1; // rstart[0] = 0 <= a
    const uint64_t r1 = pgx_sorted_upper(loc.rstart, loc.rdir, loc.rdir_shift, loc.rdir_entries, loc.n_runs, b) - 1;
    run0[i] = r0;
    n_pieces[i] = r1 - r0 + 1;
}

// one lane per (query, run) piece: start from the run's head sample, skip to the first wanted position, emit.
// seq_ids != 0 writes seqId(v) = v / max_length (r-index.hpp:429) instead of the packed position.
__global__ void __launch_bounds__(256)
pgx_locate_walk_kernel(PgxLocImage loc, const uint64_t *__restrict__ qs, const uint64_t *__restrict__ qe, uint64_t n_queries,
                       const uint64_t *__restrict__ run0, const uint64_t *__restrict__ piece_off, uint64_t n_pieces,
                       const uint64_t *__restrict__ val_off, int seq_ids, uint64_t *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pieces) return;
    // query owning piece t: last q with piece_off[q] <= t (piece_off has n_queries + 1 entries)
    uint64_t lo = 0, hi = n_queries;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (piece_off[mid + 1] <= t) lo = mid + 1; else hi = mid;
    }
    const uint64_t q = lo;
    const uint64_t run = run0[q] + (t - piece_off[q]);
    const uint64_t rs = loc.rstart[run], re = loc.rstart[run + 1]; // run = BWT[rs, re)
    const uint64_t a = qs[q] > rs ? qs[q] : rs;
    const uint64_t b = qe[q] < re - 1 ? qe[q] : re - 1;
    uint64_t v = loc.rsamp[run];
    for (uint64_t p = rs; p < a; p++) v = pgx_locate_next(loc, v); // :1280-1283
    uint64_t *dst = out + val_off[q] + (a - qs[q]);
    for (uint64_t p = a;; p++) {
        *dst++ = (seq_ids && v != PGX_NO_POSITION) ? v / loc.max_length : v;
        if (p == b) break;
        v = pgx_locate_next(loc, v);
    }
}

// ------------------------------------------------------------------------------------------
// LCE image (pgx_image.h): the suffix array in text coordinates, the text at two bits per symbol, one flag per 128-byte line of it.
// `sa` = the packed values pgx_locate_walk_kernel writes (sequence * max_length + offset, r-index.hpp:429-431).  The text is the collection as
// the index sees it: sequence q at [seq_start[q], seq_start[q] + len_q), its endmarker behind it.
__global__ void __launch_bounds__(256)
pgx_lce_seqlen_kernel(const uint64_t *__restrict__ sa, uint64_t n_seq, uint64_t max_length, unsigned long long *__restrict__ seq_len) {
    // BWT positions [0, n_seq) are the suffixes that start with an endmarker: the one of sequence q sits at offset len_q
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seq) return;
    const uint64_t v = sa[i], q = v / max_length;
    if (q < n_seq) seq_len[q] = v % max_length + 1; // (with its endmarker)
}
__global__ void __launch_bounds__(256)
pgx_lce_scatter_kernel(const uint64_t *__restrict__ sa, uint64_t n, uint64_t max_length, const uint64_t *__restrict__ seq_start, uint64_t n_seq, uint64_t c1, uint64_t c2,
                       uint64_t c3, uint64_t c4, uint64_t c5, uint32_t *__restrict__ sa32, uint8_t *__restrict__ text8, unsigned long long *__restrict__ bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = sa[i], q = v / max_length;
        const uint64_t g = q < n_seq ? seq_start[q] + v % max_length : n;
        if (g >= n) { atomicAdd(bad, 1ull); sa32[i] = 0; continue; } // (a sample outside the collection: the image is not built)
        sa32[i] = (uint32_t)g;
        // first symbol of suffix i: the C-bucket of i (\n A C G N T); codes of the packed reads: A C T G = 0 1 2 3, anything else 0xFF
        text8[g] = i < c1 ? 0xFFu : (i < c2 ? 0u : (i < c3 ? 1u : (i < c4 ? 3u : (i < c5 ? 0xFFu : 2u))));
    }
}
__global__ void __launch_bounds__(256)
pgx_lce_pack_kernel(const uint8_t *__restrict__ text8, uint64_t n, uint64_t n_words, uint32_t *__restrict__ text32, uint32_t *__restrict__ flags) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        bool special = false;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint64_t g = 16 * w + k;
            const uint32_t c = g < n ? (uint32_t)text8[g] : 0xFFu;
            special |= c > 3u;
            v |= (c & 3u) << (2u * k);
        }
        text32[w] = v;
        if (special) atomicOr(flags + (w >> 10), 1u << ((w >> 5) & 31u)); // word w lies in line w / 32 (128 bytes = 32 words); 32 lines per flag word
    }
}

// lce_lcp (pgx_image.h): entry i = the number of symbols suffix i has in common with suffix i - 1, from the 2-bit text: 64 symbols per round (five words
// of either suffix), PGX_LCP_CAP for "that many or more".  The words that were looked at lie in at most two 128-byte lines per suffix; if one of them is
// flagged (an N, an endmarker, behind the text: the 2-bit codes there mean nothing) the entry says PGX_LCP_UNKNOWN and the search compares with the text itself.
__global__ void __launch_bounds__(256)
pgx_lce_lcp_kernel(const uint32_t *__restrict__ sa32, const uint32_t *__restrict__ text32, const uint32_t *__restrict__ flags, uint64_t n, uint8_t *__restrict__ lcp) {
    typedef struct __attribute__((packed, aligned(4))) { uint32_t x, y, z, w; } u4_t;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i == 0) { lcp[0] = 0; continue; }
        const uint32_t p = sa32[i - 1], q = sa32[i];
        const uint32_t wp = p >> 4, wq = q >> 4, sp = 2u * (p & 15u), sq = 2u * (q & 15u);
        uint32_t l = PGX_LCP_CAP, rounds = 0;
#pragma unroll 1
        for (uint32_t r = 0; r < 4u && l == PGX_LCP_CAP; r++) { // 4 x 64 symbols >= the cap
            const u4_t a = *reinterpret_cast<const u4_t *>(text32 + wp + 4u * r), b = *reinterpret_cast<const u4_t *>(text32 + wq + 4u * r);
            const uint32_t a4 = text32[wp + 4u * r + 4u], b4 = text32[wq + 4u * r + 4u];
            const uint32_t A[5] = {a.x, a.y, a.z, a.w, a4}, B[5] = {b.x, b.y, b.z, b.w, b4};
            rounds = r + 1u;
#pragma unroll
            for (int u = 3; u >= 0; u--) { // (from the last unit down: the first differing one wins)
                const uint32_t df = __builtin_amdgcn_alignbit(A[u + 1], A[u], sp) ^ __builtin_amdgcn_alignbit(B[u + 1], B[u], sq);
                l = df ? 64u * r + 16u * (uint32_t)u + ((uint32_t)__builtin_ctz(df) >> 1) : l;
            }
        }
        l = l < PGX_LCP_CAP ? l : PGX_LCP_CAP;
        // the words looked at: w .. w + 4 rounds (17 at most: two lines)
        const uint32_t lp0 = wp >> 5, lp1 = (wp + 4u * rounds) >> 5, lq0 = wq >> 5, lq1 = (wq + 4u * rounds) >> 5;
        const uint32_t fl = ((flags[lp0 >> 5] >> (lp0 & 31u)) | (flags[lp1 >> 5] >> (lp1 & 31u)) | (flags[lq0 >> 5] >> (lq0 & 31u)) | (flags[lq1 >> 5] >> (lq1 & 31u))) & 1u;
        lcp[i] = (uint8_t)(fl ? PGX_LCP_UNKNOWN : l);
    }
}

// The text comparison stands for FORWARD extensions, and the FMD index answers those through the reverse complement (src/r-index.cpp:758-764): the two agree
// only where the collection holds every sequence in both orientations.  Checked here for the layout the reference's pipeline and the synthetic workloads
// write -- sequence 2 i + 1 is the reverse complement of sequence 2 i --; any position that disagrees raises the flag and the image is not built.
__global__ void __launch_bounds__(256)
pgx_lce_rc_check_kernel(const uint8_t *__restrict__ text8, const uint64_t *__restrict__ seq_start, uint64_t n_seq, uint64_t n, unsigned long long *__restrict__ bad) {
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = n_seq; // the sequence holding g: the last one whose start is <= g
        while (lo + 1 < hi) { const uint64_t mid = (lo + hi) >> 1; if (seq_start[mid] <= g) lo = mid; else hi = mid; }
        const uint64_t q = lo, t = g - seq_start[q], L = seq_start[q + 1] - seq_start[q] - 1; // (without the endmarker)
        if (t >= L) continue;
        const uint64_t p = q ^ 1ull;
        bool ok = p < n_seq && seq_start[p + 1] - seq_start[p] - 1 == L;
        if (ok) {
            const uint32_t c = text8[g], o = text8[seq_start[p] + (L - 1 - t)];
            ok = c > 3u ? o > 3u : o == (c ^ 2u); // A C T G = 0 1 2 3: the complement flips bit 1; N faces N
        }
        if (!ok) { atomicAdd(bad, 1ull); return; }
    }
}

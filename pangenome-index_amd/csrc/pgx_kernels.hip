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

#include <type_traits>

#include "pgx_device.h"

// block holding position pos (pos <= n).  One 8-byte directory entry resolves buckets with at most
// two block starts; denser buckets search the 16-bit low parts of their blocks.
template <bool LDS_IMAGE>
__device__ __forceinline__ uint32_t pgx_find_block(const PgxDevImage &img, const uint64_t *__restrict__ lds_dir,
                                                   const uint16_t *__restrict__ lds_blow, uint64_t pos) {
    const uint64_t di = pos >> img.dir_shift; // <= (n >> shift) = dir_entries - 2
    const uint64_t e = LDS_IMAGE ? lds_dir[di] : img.dir[di];
    const uint32_t lowp = (uint32_t)pos & ((1u << img.dir_shift) - 1u);
    uint32_t lo = (uint32_t)e;
    const uint32_t cnt = (uint32_t)(e >> 32) & 0xFFu;
    if (cnt <= 2) {
        const uint32_t l0 = (uint32_t)(e >> 40) & 0xFFFu, l1 = (uint32_t)(e >> 52);
        return lo - 1u + ((cnt >= 1 && l0 <= lowp) ? 1u : 0u) + ((cnt >= 2 && l1 <= lowp) ? 1u : 0u);
    }
    uint32_t hi = cnt < 255 ? lo + cnt : (uint32_t)(LDS_IMAGE ? lds_dir[di + 1] : img.dir[di + 1]);
    while (lo < hi) { // upper bound over the blocks that start inside this bucket
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t v = LDS_IMAGE ? lds_blow[mid] : img.blow[mid];
        if (v <= lowp) lo = mid + 1; else hi = mid;
    }
    return lo - 1; // block 0 starts at 0, so lo >= 1
}

// ------------------------------------------------------------------------------------------
// DENSE image (pgx_image.h): block = pos >> 6, three 64-bit planes of code bits under the same count header.
// A dense block in registers: header dwords 0..7 and plane dwords 8..13.
struct PgxDenseBlk {
    uint4 h0, h1, p01; // p01 = plane0.lo, plane0.hi, plane1.lo, plane1.hi
    uint2 p2;
};

template <bool LDS_IMAGE>
__device__ __forceinline__ PgxDenseBlk pgx_dense_load(const PgxDevImage &img, const uint4 *__restrict__ lds_blocks, uint64_t pos) {
    // in LDS the blocks are 80 bytes apart (PGX_DENSE_LDS_U4): with 64 they would start in only two bank groups
    const uint4 *bp = LDS_IMAGE ? lds_blocks + (size_t)(pos >> 6) * PGX_DENSE_LDS_U4 : img.blocks + (size_t)(pos >> 6) * 4;
    PgxDenseBlk b;
    b.h0 = bp[0]; b.h1 = bp[1]; b.p01 = bp[2];
    b.p2 = *reinterpret_cast<const uint2 *>(bp + 3);
    return b;
}

// rank sums at pos from its (loaded) block: A = count of code cv, B = sum over codes of mult[code] * count(code)
__device__ __forceinline__ void pgx_dense_rank(const PgxDenseBlk &blk, uint64_t pos, uint32_t cv, uint32_t mrow, uint64_t &A, uint64_t &B) {
    const uint4 h0 = blk.h0, h1 = blk.h1;
    uint64_t c[6];
    c[0] = (uint64_t)h0.x | ((uint64_t)(h1.z & 0xFFu) << 32);
    c[1] = (uint64_t)h0.y | ((uint64_t)((h1.z >> 8) & 0xFFu) << 32);
    c[2] = (uint64_t)h0.z | ((uint64_t)((h1.z >> 16) & 0xFFu) << 32);
    c[3] = (uint64_t)h0.w | ((uint64_t)(h1.z >> 24) << 32);
    c[4] = (uint64_t)h1.x | ((uint64_t)(h1.w & 0xFFu) << 32);
    c[5] = (uint64_t)h1.y | ((uint64_t)((h1.w >> 8) & 0xFFu) << 32);
    const uint32_t rel = (uint32_t)pos & 63u;
    // prefix mask of rel bits, as two dwords
    const uint32_t mlo = rel >= 32u ? 0xFFFFFFFFu : ((1u << rel) - 1u);
    const uint32_t mhi = rel > 32u ? ((1u << (rel - 32u)) - 1u) : 0u;
    const uint32_t a0 = blk.p01.x & mlo, a1 = blk.p01.y & mhi; // code bit 0
    const uint32_t b0 = blk.p01.z & mlo, b1 = blk.p01.w & mhi; // code bit 1
    const uint32_t d0 = blk.p2.x & mlo, d1 = blk.p2.y & mhi;   // code bit 2
    const uint32_t n1 = __popc(a0) + __popc(a1), n2 = __popc(b0) + __popc(b1), n4 = __popc(d0) + __popc(d1);
    const uint32_t n3 = __popc(a0 & b0) + __popc(a1 & b1); // code 3 = 011
    const uint32_t n5 = __popc(a0 & d0) + __popc(a1 & d1); // code 5 = 101  (codes 6, 7 never occur)
    uint32_t t[6];
    t[3] = n3; t[5] = n5;
    t[1] = n1 - n3 - n5; t[2] = n2 - n3; t[4] = n4 - n5;
    t[0] = rel - (n1 + n2 + n4 - n3 - n5);
    uint64_t a = 0, b = 0;
    uint32_t ia = 0, ib = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        a = (cv == (uint32_t)i) ? c[i] : a;
        ia = (cv == (uint32_t)i) ? t[i] : ia;
        const uint32_t w = (mrow >> (3 * i)) & 7u;
        b += c[i] * (uint64_t)w;
        ib += t[i] * w;
    }
    A = a + ia;
    B = b + ib;
}

// in-block counts of the six codes in the first (pos & 63) symbols of a dense block
__device__ __forceinline__ void pgx_dense_inblock(const PgxDenseBlk &blk, uint32_t pos_lo, uint32_t t[6]) {
    const uint32_t rel = pos_lo & 63u;
    const uint32_t mlo = rel >= 32u ? 0xFFFFFFFFu : ((1u << rel) - 1u);
    const uint32_t mhi = rel > 32u ? ((1u << (rel - 32u)) - 1u) : 0u;
    const uint32_t a0 = blk.p01.x & mlo, a1 = blk.p01.y & mhi, b0 = blk.p01.z & mlo, b1 = blk.p01.w & mhi;
    const uint32_t d0 = blk.p2.x & mlo, d1 = blk.p2.y & mhi;
    const uint32_t n1 = __popc(a0) + __popc(a1), n2 = __popc(b0) + __popc(b1), n4 = __popc(d0) + __popc(d1);
    const uint32_t n3 = __popc(a0 & b0) + __popc(a1 & b1), n5 = __popc(a0 & d0) + __popc(a1 & d1);
    t[3] = n3; t[5] = n5;
    t[1] = n1 - n3 - n5; t[2] = n2 - n3; t[4] = n4 - n5;
    t[0] = rel - (n1 + n2 + n4 - n3 - n5);
}

// Both probes of an extension in 32-bit arithmetic (BWTs shorter than 2^30: the header counts fit their low dwords):
// A0, A1 = count of code cv before p0 / p1, dB = sum over codes of mult[code] * (count before p1 - count before p0).
// One multiply per code for the pair instead of one 64-bit multiply-add per code and probe.
__device__ __forceinline__ void pgx_dense_pair32(const PgxDenseBlk &k0, uint32_t p0, const PgxDenseBlk &k1, uint32_t p1, uint32_t cv,
                                                 uint32_t mrow, uint32_t &A0, uint32_t &A1, uint32_t &dB) {
    uint32_t t0[6], t1[6];
    pgx_dense_inblock(k0, p0, t0);
    pgx_dense_inblock(k1, p1, t1);
    const uint32_t c0[6] = {k0.h0.x + t0[0], k0.h0.y + t0[1], k0.h0.z + t0[2], k0.h0.w + t0[3], k0.h1.x + t0[4], k0.h1.y + t0[5]};
    const uint32_t c1[6] = {k1.h0.x + t1[0], k1.h0.y + t1[1], k1.h0.z + t1[2], k1.h0.w + t1[3], k1.h1.x + t1[4], k1.h1.y + t1[5]};
    uint32_t a0 = 0, a1 = 0, d = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        a0 = (cv == (uint32_t)i) ? c0[i] : a0;
        a1 = (cv == (uint32_t)i) ? c1[i] : a1;
        d += (c1[i] - c0[i]) * ((mrow >> (3 * i)) & 7u);
    }
    A0 = a0; A1 = a1; dB = d;
}

// ------------------------------------------------------------------------------------------
// DENSE2 image (pgx_image.h): 384 symbols per 128-byte block = 32-byte header + three 32-byte sub-blocks of 128 symbols in two
// bit planes, exception runs for \n and N; positions below 2^32.  A probe loads the header and one sub-block.
struct PgxDense2Blk {
    uint4 h0, h1; // header: counts A C G T | N, exceptions, sub-block counts (64 bits)
    uint4 p0, p1; // the sub-block of the probe: plane 0, plane 1
};
__device__ __forceinline__ PgxDense2Blk pgx_dense2_load(const PgxDevImage &img, uint32_t pos, uint32_t &rel) {
    const uint32_t blk = (uint32_t)(((uint64_t)pos * 0xAAAAAAABull) >> 40); // pos / 384
    rel = pos - blk * PGX_D2_SYMS;
    const uint4 *bp = img.blocks + (size_t)blk * 8;
    PgxDense2Blk b;
    b.h0 = bp[0]; b.h1 = bp[1];
    const uint4 *sp = bp + 2 + 2 * (rel >> 7);
    b.p0 = sp[0]; b.p1 = sp[1];
    return b;
}
// counts of the six nuc codes (\n A C G N T) in BWT[0, pos) for the probe whose block / sub-block was loaded
__device__ __forceinline__ void pgx_dense2_counts(const PgxDevImage &img, const PgxDense2Blk &b, uint32_t pos, uint32_t rel, uint32_t c[6]) {
    const uint32_t sub = rel >> 7, r = rel & 127u;
    // in-block counts before the sub-block (three 9-bit fields per sub-block boundary)
    const uint64_t sc = ((uint64_t)b.h1.z | ((uint64_t)b.h1.w << 32)) >> (sub == 2u ? 27 : 0);
    uint32_t n0 = sub ? (uint32_t)sc & 511u : 0u, n1 = sub ? (uint32_t)(sc >> 9) & 511u : 0u, n3 = sub ? (uint32_t)(sc >> 18) & 511u : 0u;
    const uint32_t a[4] = {b.p0.x, b.p0.y, b.p0.z, b.p0.w}, d[4] = {b.p1.x, b.p1.y, b.p1.z, b.p1.w};
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const int32_t t = (int32_t)r - 32 * h; // bits of this dword that lie below the position
        const uint32_t m = t >= 32 ? 0xFFFFFFFFu : (t > 0 ? ((1u << t) - 1u) : 0u);
        const uint32_t x = a[h] & m, y = d[h] & m;
        n0 += __popc(x); n1 += __popc(y); n3 += __popc(x & y);
    }
    uint32_t e0 = 0, e4 = 0;
    const uint32_t ec = b.h1.y >> 24;
    if (ec) { // rare: the block holds endmarkers or N
        const uint32_t *ep = img.exc + (b.h1.y & 0xFFFFFFu);
        for (uint32_t i = 0; i < ec; i++) {
            const uint32_t u = ep[i], st = u & 511u, ln = (u >> 9) & 511u;
            const uint32_t cnt = rel > st ? min(rel - st, ln) : 0u;
            if ((u >> 18) & 1u) e4 += cnt; else e0 += cnt;
        }
    }
    const uint32_t hsum = b.h0.x + b.h0.y + b.h0.z + b.h0.w + b.h1.x;
    c[0] = (pos - rel) - hsum + e0;               // \n: block start minus the five stored counts
    c[1] = b.h0.x + rel - (n0 + n1 - n3) - e0 - e4; // A
    c[2] = b.h0.y + n0 - n3;                      // C
    c[3] = b.h0.z + n1 - n3;                      // G
    c[4] = b.h1.x + e4;                           // N
    c[5] = b.h0.w + n3;                           // T
}
// both probes of an extension: A0, A1 = count of code cv before p0 / p1; dB = sum over codes of mult[code] * (count before p1 -
// count before p0).  NARROW: modulo 2^32 (like pgx_dense_pair32); otherwise modulo 2^64 from the exact 32-bit counts.
template <bool NARROW>
__device__ __forceinline__ void pgx_dense2_pair(const PgxDevImage &img, uint32_t p0, uint32_t p1, uint32_t cv, uint32_t mrow, uint64_t &A0, uint64_t &A1,
                                                uint64_t &dB, bool lower = true) {
    uint32_t r0, r1;
    const PgxDense2Blk k0 = pgx_dense2_load(img, p0, r0), k1 = pgx_dense2_load(img, p1, r1);
    if (lower) __builtin_amdgcn_s_setprio(0); // (the find_mems kernels raise their priority on the way to the loads: see pgx_find_mems_pairs_kernel)
    uint32_t c0[6], c1[6];
    pgx_dense2_counts(img, k0, p0, r0, c0);
    pgx_dense2_counts(img, k1, p1, r1, c1);
    uint32_t a0 = 0, a1 = 0, d32 = 0;
    uint64_t d64 = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        a0 = (cv == (uint32_t)i) ? c0[i] : a0;
        a1 = (cv == (uint32_t)i) ? c1[i] : a1;
        const uint32_t w = (mrow >> (3 * i)) & 7u;
        if (NARROW) d32 += (c1[i] - c0[i]) * w;
        else d64 += (uint64_t)((int64_t)c1[i] - (int64_t)c0[i]) * (uint64_t)w;
    }
    A0 = a0; A1 = a1; dB = NARROW ? (uint64_t)d32 : d64;
}
// one probe (primitives)
__device__ __forceinline__ void pgx_dense2_rank(const PgxDevImage &img, uint32_t pos, uint32_t cv, uint32_t mrow, uint64_t &A, uint64_t &B) {
    uint32_t rel;
    const PgxDense2Blk k = pgx_dense2_load(img, pos, rel);
    uint32_t c[6];
    pgx_dense2_counts(img, k, pos, rel, c);
    uint64_t a = 0, bb = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        a = (cv == (uint32_t)i) ? (uint64_t)c[i] : a;
        bb += (uint64_t)c[i] * (uint64_t)((mrow >> (3 * i)) & 7u);
    }
    A = a; B = bb;
}

// ------------------------------------------------------------------------------------------
// WIDE DENSE2 (pgx_image.h): the same blocks, header counts as deltas against the 64-bit bases of the block's superblock; positions
// and counts in 64 bits.  `sb` = the base table (img.sbase2 or its LDS copy): 8 words per superblock {A, C, G, T, N, their sum}.
// (MULHI: the form of pos / 384 this function had until round 3, kept for scripts/anomaly_probe.py only)
template <bool MULHI = false>
__device__ __forceinline__ PgxDense2Blk pgx_dense2w_load(const PgxDevImage &img, uint64_t pos, uint32_t &rel, uint32_t &blk) {
    blk = MULHI ? (uint32_t)(__umul64hi(pos, 0xAAAAAAAAAAAAAAABull) >> 8)
                : (uint32_t)(((pos >> 7) * 0xAAAAAAABull) >> 33); // pos / 384 = (pos / 128) / 3, exact while pos / 128 < 2^32
    rel = (uint32_t)(pos - (uint64_t)blk * PGX_D2_SYMS);
    const uint4 *bp = img.blocks + (size_t)blk * 8;
    PgxDense2Blk b;
    b.h0 = bp[0]; b.h1 = bp[1];
    const uint4 *sp = bp + 2 + 2 * (rel >> 7);
    b.p0 = sp[0]; b.p1 = sp[1];
    return b;
}
__device__ __forceinline__ void pgx_dense2w_counts(const PgxDevImage &img, const uint64_t *__restrict__ sb, const PgxDense2Blk &b, uint64_t pos, uint32_t rel,
                                                   uint32_t blk, uint64_t c[6]) {
    uint32_t d[6];
    pgx_dense2_counts(img, b, rel, rel, d); // with pos = rel: d[0] = -(sum of the five deltas) + e0, the others delta + in-block count
    const uint64_t *base = sb + (size_t)(blk >> img.d2_sb_shift) * 8;
    c[0] = (pos - rel) - base[5] + (uint64_t)(int64_t)(int32_t)d[0]; // (the deltas of a superblock stay below 2^31: d[0] is a small negative number)
    c[1] = base[0] + d[1];
    c[2] = base[1] + d[2];
    c[3] = base[2] + d[3];
    c[4] = base[4] + d[4];
    c[5] = base[3] + d[5];
}
__device__ __forceinline__ void pgx_dense2w_pair(const PgxDevImage &img, const uint64_t *__restrict__ sb, uint64_t p0, uint64_t p1, uint32_t cv, uint32_t mrow,
                                                 uint64_t &A0, uint64_t &A1, uint64_t &dB, bool lower = true) {
    uint32_t r0, r1, b0, b1;
    const PgxDense2Blk k0 = pgx_dense2w_load(img, p0, r0, b0), k1 = pgx_dense2w_load(img, p1, r1, b1);
    if (lower) __builtin_amdgcn_s_setprio(0);
    uint64_t c0[6], c1[6];
    pgx_dense2w_counts(img, sb, k0, p0, r0, b0, c0);
    pgx_dense2w_counts(img, sb, k1, p1, r1, b1, c1);
    uint64_t a0 = 0, a1 = 0, d = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        a0 = (cv == (uint32_t)i) ? c0[i] : a0;
        a1 = (cv == (uint32_t)i) ? c1[i] : a1;
        d += (c1[i] - c0[i]) * (uint64_t)((mrow >> (3 * i)) & 7u);
    }
    A0 = a0; A1 = a1; dB = d;
}
template <bool MULHI = false>
__device__ __forceinline__ void pgx_dense2w_rank(const PgxDevImage &img, uint64_t pos, uint32_t cv, uint32_t mrow, uint64_t &A, uint64_t &B) {
    uint32_t rel, blk;
    const PgxDense2Blk k = pgx_dense2w_load<MULHI>(img, pos, rel, blk);
    uint64_t c[6];
    pgx_dense2w_counts(img, img.sbase2, k, pos, rel, blk, c);
    uint64_t a = 0, bb = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        a = (cv == (uint32_t)i) ? c[i] : a;
        bb += c[i] * (uint64_t)((mrow >> (3 * i)) & 7u);
    }
    A = a; B = bb;
}

// ------------------------------------------------------------------------------------------
// rank probe: A = count of code `cv` in BWT[0,pos), B = sum over codes of mult[code] * count(code)
// (both modulo 2^64; only differences of two probes are ever used).
template <bool LDS_IMAGE>
__device__ __forceinline__ void pgx_rank_ab(const PgxDevImage &img, const uint4 *__restrict__ lds_blocks,
                                            const uint64_t *__restrict__ lds_dir, const uint16_t *__restrict__ lds_blow,
                                            uint64_t pos, uint32_t cv, uint32_t mrow, uint64_t &A, uint64_t &B) {
    if (pos > img.n) pos = img.n; // predecessor(pos >= size) = last block, rel past the end = totals
    if (img.dense == 3) { pgx_dense2w_rank(img, pos, cv, mrow, A, B); return; }
    if (img.dense == 2) { pgx_dense2_rank(img, (uint32_t)pos, cv, mrow, A, B); return; }
    if (img.dense) {
        pgx_dense_rank(pgx_dense_load<LDS_IMAGE>(img, lds_blocks, pos), pos, cv, mrow, A, B);
        return;
    }
    const uint32_t lo = pgx_find_block<LDS_IMAGE>(img, lds_dir, lds_blow, pos);
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
    uint32_t rel = (uint32_t)(pos - start); // block extent <= 16 * 4095
    uint32_t ia = 0, ib = 0;
    const uint32_t arow = 1u << (3 * cv);   // one-hot weight row selecting code cv
    const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int e = 0; e < PGX_BLOCK_RUNS; e++) {
        const uint32_t w = rw[e >> 1];
        const uint32_t sh = (e & 1) ? (w >> 28) : __builtin_amdgcn_ubfe(w, 12, 4);          // 3 * code
        const uint32_t len = (e & 1) ? __builtin_amdgcn_ubfe(w, 16, 12) : (w & PGX_RUN_LEN_MAX);
        const uint32_t take = min(len, rel);
        rel -= take;
        ia += take * __builtin_amdgcn_ubfe(arow, sh, 3);
        ib += take * __builtin_amdgcn_ubfe(mrow, sh, 3);
    }
    A = a + ia;
    B = b + ib;
}

// One block decode ("trip") of the rank machinery: decodes the block holding p and returns
//   Ap, Bp  rank sums at p (primary position; p <= n)
//   As, Bs  rank sums at the secondary position p1 when `with_secondary` and this block also serves p1
//           (`covered`): once an interval is narrow (s ~ number of haplotypes) both probes of an extension
//           fall into the same 64-byte block and one decode answers both.
// A = count of code cv, B = sum over codes of mult[code] * count(code), both modulo 2^64.
template <bool LDS_IMAGE>
__device__ __forceinline__ void pgx_probe(const PgxDevImage &img, const uint4 *__restrict__ lds_blocks,
                                          const uint64_t *__restrict__ lds_dir, const uint16_t *__restrict__ lds_blow,
                                          uint64_t p, uint64_t p1, bool with_secondary, uint32_t cv, uint32_t mrow,
                                          uint64_t &Ap, uint64_t &Bp, uint64_t &As, uint64_t &Bs, bool &covered) {
    const uint32_t lo = pgx_find_block<LDS_IMAGE>(img, lds_dir, lds_blow, p);
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
    uint32_t relp = (uint32_t)(p - start);   // primary position (inside the block)
    const uint64_t d1 = p1 - start;          // p1 relative to this block; wraps when p1 < start
    uint32_t rels = with_secondary ? (d1 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d1) : 0u;
    uint32_t iap = 0, ibp = 0, ias = 0, ibs = 0, total = 0;
    const uint32_t arow = 1u << (3 * cv); // one-hot weight row selecting code cv
    const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int e = 0; e < PGX_BLOCK_RUNS; e++) {
        // entry = (3 * code) << 12 | len: the stored shift indexes the 3-bit weight rows directly
        const uint32_t w = rw[e >> 1];
        const uint32_t sh = (e & 1) ? (w >> 28) : __builtin_amdgcn_ubfe(w, 12, 4);
        const uint32_t len = (e & 1) ? __builtin_amdgcn_ubfe(w, 16, 12) : (w & PGX_RUN_LEN_MAX);
        const uint32_t tp = min(len, relp), ts = min(len, rels);
        const uint32_t wa = __builtin_amdgcn_ubfe(arow, sh, 3), wm = __builtin_amdgcn_ubfe(mrow, sh, 3);
        total += len;
        relp -= tp;
        rels -= ts;
        iap += tp * wa;
        ias += ts * wa;
        ibp += tp * wm;
        ibs += ts * wm;
    }
    Ap = a + iap;
    Bp = b + ibp;
    As = a + ias;
    Bs = b + ibs;
    // p1 is served by this block when it lies strictly inside it (a probe AT the block end belongs to the
    // next block, whose header may carry a different quirk value), or at the end of the BWT
    covered = with_secondary && (d1 < (uint64_t)total || (rels == 0 && lo + 1 == img.n_blocks));
}

// The two rank probes of one extension, rank(pos0) and rank(pos1) with pos1 = pos0 + s, as at most two
// trips of pgx_probe in a rolled loop (the decode exists once in the instruction stream).
// Outputs A0, A1 and B1 - B0.  (Used by the primitives; the find_mems kernel schedules trips itself.)
template <bool LDS_IMAGE, bool MAYBE_DENSE = true>
__device__ __forceinline__ void pgx_rank_pair(const PgxDevImage &img, const uint4 *__restrict__ lds_blocks,
                                              const uint64_t *__restrict__ lds_dir, const uint16_t *__restrict__ lds_blow,
                                              uint64_t pos0, uint64_t pos1, uint32_t cv, uint32_t mrow, uint64_t &A0,
                                              uint64_t &A1, uint64_t &dB) {
    const uint64_t p0 = pos0 > img.n ? img.n : pos0, p1 = pos1 > img.n ? img.n : pos1;
    uint64_t B0 = 0, B1 = 0;
    if (MAYBE_DENSE && img.dense == 3) { pgx_dense2w_pair(img, img.sbase2, p0, p1, cv, mrow, A0, A1, dB); return; }
    if (MAYBE_DENSE && img.dense == 2) { pgx_dense2_pair<false>(img, (uint32_t)p0, (uint32_t)p1, cv, mrow, A0, A1, dB); return; }
    if (MAYBE_DENSE && img.dense) { // two independent block loads, no directory
        const PgxDenseBlk k0 = pgx_dense_load<LDS_IMAGE>(img, lds_blocks, p0), k1 = pgx_dense_load<LDS_IMAGE>(img, lds_blocks, p1);
        pgx_dense_rank(k0, p0, cv, mrow, A0, B0);
        pgx_dense_rank(k1, p1, cv, mrow, A1, B1);
        dB = B1 - B0;
        return;
    }
    A0 = 0; A1 = 0;
    bool done = false;
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        if (!done) {
            uint64_t Ap, Bp, As, Bs;
            bool covered;
            pgx_probe<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_blow, it ? p1 : p0, p1, it == 0, cv, mrow, Ap, Bp, As, Bs, covered);
            if (it == 0) {
                A0 = Ap; B0 = Bp;
                if (covered) { A1 = As; B1 = Bs; done = true; }
            } else {
                A1 = Ap; B1 = Bp;
            }
        }
    }
    dB = B1 - B0;
}

// one FMD extension of (k, kp, s) by `byte` (backward, or forward = backward on the swapped
// interval by the complement, folded into ext_tab[256 + byte]).  Returns the new size (0 = empty).
template <bool LDS_IMAGE>
__device__ __forceinline__ void pgx_extend(const PgxDevImage &img, const uint4 *lds_blocks, const uint64_t *lds_dir,
                                           const uint16_t *lds_blow, const uint32_t *s_ext, const uint64_t *s_C,
                                           uint64_t &k, uint64_t &kp, uint64_t &s, uint32_t byte, bool fwd) {
    const uint32_t e = s_ext[(fwd ? 256u : 0u) + byte];
    const uint32_t cv = PGX_EXT_CV(e), mrow = PGX_EXT_M(e);
    const uint64_t kk = fwd ? kp : k, kq = fwd ? k : kp;
    uint64_t A1, A0, dB;
    pgx_rank_pair<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_blow, kk, kk + s, cv, mrow, A0, A1, dB);
    if (PGX_EXT_KILL(e) || A0 >= A1) { // rank_k >= rank_ks -> bi_interval(0,0,0), src/r-index.cpp:751
        k = 0; kp = 0; s = 0;
        return;
    }
    const uint64_t nk = A0 + s_C[PGX_EXT_V(e)], nq = kq + dB;
    s = A1 - A0;
    k = fwd ? nq : nk;
    kp = fwd ? nk : nq;
}

template <bool LDS_IMAGE>
__device__ __forceinline__ void pgx_stage_tables(const PgxDevImage &img, uint32_t *s_ext, uint64_t *s_C, uint4 *lds_blocks,
                                                 uint64_t *lds_dir, uint16_t *lds_blow) {
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) s_ext[i] = img.consts->ext_tab[i];
    if (threadIdx.x < 8) s_C[threadIdx.x] = img.consts->C[threadIdx.x];
    if (LDS_IMAGE) {
        const uint32_t nb4 = img.n_blocks * 4;
        for (uint32_t i = threadIdx.x; i < nb4; i += blockDim.x) lds_blocks[img.dense ? (i >> 2) * PGX_DENSE_LDS_U4 + (i & 3u) : i] = img.blocks[i];
        if (!img.dense)
            for (uint64_t i = threadIdx.x; i < img.dir_entries; i += blockDim.x) lds_dir[i] = img.dir[i];
        if (!img.dense)
            for (uint32_t i = threadIdx.x; i < img.n_blocks; i += blockDim.x) lds_blow[i] = img.blow[i];
    }
    __syncthreads();
}

// dynamic LDS carve (16-byte aligned base): [blocks | dir | blow]
#define PGX_LDS_CARVE(img)                                                                   \
    extern __shared__ __align__(16) unsigned char pgx_dyn_lds[];                              \
    uint4 *lds_blocks = reinterpret_cast<uint4 *>(pgx_dyn_lds);                               \
    uint64_t *lds_dir = reinterpret_cast<uint64_t *>(pgx_dyn_lds + (size_t)(img).n_blocks * PGX_BLOCK_BYTES);    \
    uint16_t *lds_blow = reinterpret_cast<uint16_t *>(pgx_dyn_lds + (size_t)(img).n_blocks * PGX_BLOCK_BYTES + (img).dir_entries * 8)

// ------------------------------------------------------------------------------------------
// k-mer seeds.  A backward stage of find_mems_function that starts from the full interval (step 1 at j = x + min_len - 1, step 3
// at j = e) performs its first K extensions over the window P[j - K + 1 .. j], last byte first; the table holds the result of
// those K extensions for every ACGT window, computed on the device by the same pgx_extend (so every quirk the tables carry is in
// it), and for windows that leave the index the number of extensions until the interval became empty.  One 16-byte load then
// replaces K extensions = up to 2 K line fetches, the widest ones of the search; the counters advance by K (or by the death
// depth), so MEMs, returned start positions and n_extensions stay those of the stepwise search.
//   index = sum over window bytes b_i (memory order) of code(b_i) << 2 i, code = (byte >> 1) & 3: A 0, C 1, T 2, G 3
__device__ __forceinline__ uint32_t pgx_seed_codes(uint64_t x, uint64_t &bad) {
    const uint64_t c = (x >> 1) & 0x0303030303030303ull;
    const uint64_t b0 = c & 0x0101010101010101ull, b1 = (c >> 1) & 0x0101010101010101ull;
    // the byte each code stands for; anything else in the window (N, lower case, \0, ...) makes it unusable
    const uint64_t recon = 0x4141414141414141ull + 2 * (b0 & ~b1) + 0x13 * (b1 & ~b0) + 6 * (b0 & b1);
    bad = x ^ recon;
    uint64_t t = (c | (c >> 6)) & 0x000F000F000F000Full;
    t = (t | (t >> 12)) & 0x000000FF000000FFull;
    t = (t | (t >> 24)) & 0xFFFFull;
    return (uint32_t)t;
}
__device__ __forceinline__ bool pgx_seed_index(uint64_t lo, uint64_t hi, uint32_t K, uint32_t &idx) {
    uint64_t badlo, badhi;
    const uint32_t ilo = pgx_seed_codes(lo, badlo), ihi = pgx_seed_codes(hi, badhi);
    const uint32_t nlo = K < 8u ? K : 8u, nhi = K > 8u ? K - 8u : 0u;
    const uint64_t mlo = nlo == 8u ? ~0ull : ((1ull << (8u * nlo)) - 1ull), mhi = nhi == 8u ? ~0ull : ((1ull << (8u * nhi)) - 1ull);
    idx = (ilo & ((1u << (2u * nlo)) - 1u)) | ((ihi & ((1u << (2u * nhi)) - 1u)) << 16);
    return ((badlo & mlo) | (badhi & mhi)) == 0ull;
}

// level `level` (4^level entries, src; level 0 = the full interval) -> level + 1: entry (p << 2 | c) = entry p extended by code c
__global__ void __launch_bounds__(256)
pgx_seed_build_kernel(PgxDevImage img, const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t level, uint64_t n_dst, uint64_t limit, int end_table) {
    __shared__ uint32_t s_ext[512];
    __shared__ uint64_t s_C[8];
    pgx_stage_tables<false>(img, s_ext, s_C, nullptr, nullptr, nullptr);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dst; i += (uint64_t)gridDim.x * blockDim.x) { // (a level can have 2^32 entries)
    uint64_t k = 0, kp = 0, s = img.n;
    uint32_t depth = 0;
    const uint32_t base_depth = end_table ? 1u : 0u; // the end table: level 0 is the full interval extended by 0 (pattern[len]), which counts as an extension
    if (end_table && !level) {
        pgx_extend<false>(img, nullptr, nullptr, nullptr, s_ext, s_C, k, kp, s, 0u, false);
        if (s == 0) { k = 0; kp = 0; depth = 1u; }
    }
    if (level) {
        const uint4 e = src[i >> 2];
        k = (uint64_t)e.x | ((uint64_t)(e.w & 0xFFu) << 32);
        kp = (uint64_t)e.y | ((uint64_t)((e.w >> 8) & 0xFFu) << 32);
        s = (uint64_t)e.z | ((uint64_t)((e.w >> 16) & 0xFFu) << 32);
        depth = e.w >> 24;
    }
    if (s != 0) {
        const uint32_t byte = (0x47544341u >> (8u * (uint32_t)(i & 3))) & 0xFFu; // "ACTG"[code]
        pgx_extend<false>(img, nullptr, nullptr, nullptr, s_ext, s_C, k, kp, s, byte, false);
        if (s == 0) { k = 0; kp = 0; depth = level + 1 + base_depth; }
        else if (k >= limit || kp >= limit || k + s >= limit || kp + s >= limit || k + s < k || kp + s < kp) { k = 0; kp = 0; s = 0; depth = PGX_SEED_UNUSABLE; }
    }
    uint4 o;
    o.x = (uint32_t)k; o.y = (uint32_t)kp; o.z = (uint32_t)s;
    o.w = (uint32_t)(k >> 32) | ((uint32_t)(kp >> 32) << 8) | ((uint32_t)(s >> 32) << 16) | (depth << 24);
    dst[i] = o;
  }
}

// MEM slots of one chunk of reads: the first PGX_FAST_SLOTS MEMs of a read live in a dense array at the start of the slot buffer, slot-major (the
// k-th MEM of read r of the chunk: entry k * chunk_reads + r) -- a read has 1.9 MEMs on average, so the compaction reads the 0.6 GB that exist of
// that 1.28 GB array, coalesced over neighbouring reads (read-major, entries 4 r .. 4 r + 3, it read every line: 0.53 -> 0.41 ms at chr22 scale) --,
// further ones in the arena or at the read's worst-case offset (slot_off) behind that array
#define PGX_FAST_SLOTS 4u
#ifndef PGX_PAIRS_PACKED_WAVES
#define PGX_PAIRS_PACKED_WAVES 5 // waves per SIMD the packed narrow pairs kernel is compiled for (95 VGPRs with the inline dense2 step; 79 and six waves without it were no faster)
#endif
#define PGX_PK_GROUP 12u // packed words of a read fetched per round of loads when a lane takes the read
__device__ __forceinline__ uint64_t pgx_slot_index(uint64_t read_in_chunk, uint64_t chunk_reads, uint64_t slot, uint32_t nm) {
    return nm < PGX_FAST_SLOTS ? (uint64_t)nm * chunk_reads + read_in_chunk : chunk_reads * PGX_FAST_SLOTS + slot + nm;
}
// Where the fifth and later MEMs of a read go (`slot` of pgx_slot_index).  Worst-case layout (ovf_cap == 0): the read's offset in the scan of
// min(len, len - min_len + 1), 131 slots per 150-bp read -- 42 GB for 10 M reads that write 0.6 GB.  ARENA (round 3): a read reserves its extent when it
// emits its fifth MEM (2 % of the reads do) -- as many slots as it has start positions left, which bounds what it can still emit -- with one atomic,
// and records it in ovf_base[rid] for its later MEMs, the kernels that continue the read, and the compaction.  The arena is PGX_ARENA_SUBS sub-arenas,
// one per residue of the read number, each with a counter on a cache line of its own (ctr + PGX_CTR_ARENA0 + 16 sub): one counter for everybody
// serialised 48 k atomics into 0.7 ms on the x fixture (a 0.6 ms kernel).  A sub-arena that proves too small raises PGX_CTR_OVF_ABORT (the writes
// then land at its start, in bounds) and the host repeats the chunk in the worst-case layout.
__device__ __forceinline__ uint64_t pgx_slot_extent(const uint64_t *__restrict__ slot_off, uint64_t slot_base, uint32_t *__restrict__ ovf_base, uint64_t ovf_cap,
                                                    unsigned long long *__restrict__ ctr, uint64_t rid, uint32_t nm, int32_t len, int32_t x, uint64_t min_len) {
    if (!ovf_cap) return slot_off[rid] - slot_base;
    if (nm == PGX_FAST_SLOTS) {
        const int64_t ml = min_len ? (int64_t)min_len : 1;
        const int64_t left = (int64_t)len - ml - (int64_t)x + 1; // start positions from x on (x itself has just produced a MEM)
        const unsigned long long ext = left > 0 ? (unsigned long long)left : 1ull;
        // (by read id, not by workgroup: what a sub-arena is asked for then does not depend on which workgroup took which reads, so a run sized from
        //  the one before fits -- reads that need many slots come in clusters, e.g. the reads cut from an N run, and landed in a few sub-arenas)
        const uint32_t sub = (uint32_t)rid & (PGX_ARENA_SUBS - 1u);
        const uint64_t sub_cap = ovf_cap / PGX_ARENA_SUBS;
        unsigned long long at = atomicAdd(ctr + PGX_CTR_ARENA0 + 16u * sub, ext);
        if (at + ext > sub_cap) { ctr[PGX_CTR_OVF_ABORT] = 1ull; at = 0ull; }
        at += (unsigned long long)sub * sub_cap;
        ovf_base[rid] = (uint32_t)at;
        return at - PGX_FAST_SLOTS; // (pgx_slot_index adds nm)
    }
    return (uint64_t)ovf_base[rid] - PGX_FAST_SLOTS;
}

// ------------------------------------------------------------------------------------------
// find_all_mems for a batch.  State machine of find_mems_function (algorithm.hpp:653-736):
//   phase 1  backward from j = x+min_len-1 down to x          (:666-676)
//   phase 2  forward  from j = x+min_len   up to len-1        (:684-696)  -> emit MEM (:713)
//   phase 3  backward from j = e down to x+1, fresh interval  (:718-735); pattern[len] reads 0
//
// Persistent work-queue kernel: one lane owns one live read; every trip of the main loop performs
// exactly one extension for every live lane.  A lane whose read is finished is refilled at once
// (reads differ 2-3x in their extension counts, so a static read->lane map leaves most lanes idle):
// the wavefront keeps a private reservoir [rnext, rend) of read ids that lane 0 replenishes with
// one atomicAdd of PGX_FM_BATCH on the global cursor, and idle lanes take ids from it in lane order
// (ballot + prefix popcount).  Every wave leaves the loop once the cursor has passed n_reads and
// all its lanes are idle.  MEMs go to per-read slots, so the output does not depend on scheduling.
//
// Heavy reads: a read whose suffix ends a sequence makes step 3 walk the whole read for every start position
// (pattern[len] = 0 is the endmarker): ~len^2 / 2 extensions in one dependent chain, 100 x an
// ordinary read, tens of milliseconds for one lane.  A lane that has spent `heavy_ext` extensions on its read hands the
// rest (rid, next start, MEMs so far) to pgx_find_mems_heavy_kernel at the next start-position boundary.
#define PGX_FM_BATCH (LDS_IMAGE ? 128u : 32u) // reads per grab: fewer leave less in the wave's reserve when the queue runs dry (synth: 32 < 64 < 128 < 256),
                                               // but the LDS kernels are fast enough to feel the contention on the cursor (x: 128 < 64)
// NARROW (dense images of BWTs shorter than 2^30 only): interval coordinates and rank sums in 32 bits -- half the moves,
// selects and adds of the loop.  Sound because every true value is < 2^32 there; the junk coordinates the COMPAT quirks can
// produce are caught at the two additions that could wrap (counter slot 9 is raised and the host repeats the chunk in 64 bits).
template <bool LDS_IMAGE, int DENSE, bool NARROW, bool SEED>
__global__ void __launch_bounds__(PGX_FM_THREADS, DENSE == 0 ? 3 : PGX_FM_WAVES_PER_SIMD) // the run-length decode does not fit 128 VGPRs without spilling
pgx_find_mems_kernel(PgxDevImage img, const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets,
                     uint64_t n_reads, uint64_t min_len, uint64_t min_occ, const uint64_t *__restrict__ slot_off,
                     pgx_mem *__restrict__ slots, uint32_t *__restrict__ mem_count, unsigned long long *__restrict__ n_ext_total,
                     unsigned long long *__restrict__ cursor, uint64_t first_read, uint64_t slot_base, uint32_t heavy_ext, uint32_t heavy_cap,
                     pgx_heavy_item *__restrict__ heavy_list, unsigned long long *__restrict__ heavy_count,
                     const pgx_heavy_item *__restrict__ rid_list, const unsigned long long *__restrict__ rid_count, uint32_t *__restrict__ ovf_base, uint64_t ovf_cap) {
    __shared__ uint32_t s_ext[512];
    __shared__ uint64_t s_C[8];
    __shared__ uint64_t s_sb[DENSE == 3 ? PGX_SB_MAX * 8 : 1]; // WIDE dense2: superblock bases
    PGX_LDS_CARVE(img);
    if (DENSE == 3) for (uint32_t i = threadIdx.x; i < img.n_sb2 * 8u; i += blockDim.x) s_sb[i] = img.sbase2[i];
    pgx_stage_tables<LDS_IMAGE>(img, s_ext, s_C, lds_blocks, lds_dir, lds_blow);
    // rid_list (may be NULL): the launch serves the reads listed there (the reads with a byte outside A C G T, which the pairs kernel skips: the launch
    // on the second stream), *rid_count of them, each from the start position listed, keeping the MEMs written before
    const uint64_t chunk_first = first_read, chunk_reads = n_reads - first_read; // (n_reads is the END of the chunk)
    if (rid_list) { first_read = 0; n_reads = *rid_count; }

    static_assert(!NARROW || DENSE, "the 32-bit state exists for the dense image only");
    static_assert(!SEED || DENSE, "k-mer seeds exist for the dense images");
    static_assert(DENSE < 2 || !LDS_IMAGE, "the dense2 image is never staged in LDS");
    static_assert(DENSE != 3 || !NARROW, "the wide dense2 image is walked in 64 bits");
    typedef typename std::conditional<NARROW, uint32_t, uint64_t>::type pos_t;
    const int lane = threadIdx.x & 63;
    const pos_t n = (pos_t)img.n;
    uint64_t rid = 0, base = 0;
    int32_t len = 0, x = 0, j = 0;
    pos_t k = 0, kp = 0, s = 0, Jk = 0, Js = 0;
    uint32_t nm = 0, next = 0, next0 = 0; // next0: value of `next` when the current read was taken
    int ph = 0;                      // 0 = idle (no read, or read finished)
    // read bytes cached in registers: 32 (absolute, 32-aligned offset) when the image is in global memory and the loop waits
    // on memory anyway (8 -> 16 -> 32 bytes: 3.94 -> 3.77 -> 3.72 ms on the synthetic pangenome), 8 when it is in LDS and the
    // loop is bound by issue slots.  The reads buffer is padded with 32 zero bytes, so the window never overruns.
    uint64_t win = 0, win_hi = 0, win2 = 0, win3 = 0, win_at = ~0ull;
    pos_t A0 = 0, B0 = 0;             // first-probe sums of an extension whose second probe is pending
    bool pend = false;
    uint32_t fresh = 0;              // (a 32-bit flag: as a bool captured by the lambdas below it ended up in scratch memory) SEED: the interval is the full one and a backward stage is about to start (the seed table may apply)
    bool ovf = false;                // NARROW: some addition left 32 bits (reported once, when the wave leaves)
    uint64_t rnext = 0, rend = 0;    // wave-uniform reservoir of read ids
    bool exhausted = false;          // wave-uniform: the global cursor has passed n_reads
    unsigned long long ln_blk = 0, ln_seed = 0; // wave-uniform (scalar registers): image lines / seed entries the wave asked for (PGX_CTR_FM_LINES / _SEEDS; images in global memory)
#ifdef PGX_FM_STATS
    unsigned long long st_trips = 0, st_live = 0; // diagnostics build only (scripts/fm_stats.sh)
#endif

    // begin(x): entry of find_mems_function; finishing a read records its MEM count
    auto begin = [&]() __attribute__((always_inline)) {
        if (x >= len || (uint64_t)(len - x) < min_len) { ph = 0; mem_count[rid] = nm; return; } // :745 / :658
        if (heavy_ext && next - next0 >= heavy_ext && len <= (int32_t)PGX_FM_HEAVY_MAXLEN) { // hand the rest of a heavy read on
            const unsigned long long at = atomicAdd(heavy_count, 1ull);
            if (at < (unsigned long long)heavy_cap) {
                pgx_heavy_item it;
                it.rid = rid; it.x = (uint32_t)x; it.nm = nm;
                heavy_list[at] = it;
                ph = 0;
                return;
            }
        }
        k = 0; kp = 0; s = n;
        if (min_len == 0) { // step 1 runs zero times (:666); step 2 starts at j = x
            Jk = 0; Js = n; j = x; ph = 2;
        } else {
            j = x + (int32_t)min_len - 1; ph = 1;
            fresh = 1u;
        }
    };
    // the state machine below funnels every "next start position" through one begin() (the lambda is inlined per call site)
    uint32_t restart = 0;
    // emit the MEM [x, e) and set up step 3
    auto emit = [&]() __attribute__((always_inline)) {
        pgx_mem m;
        m.start = (uint64_t)x; m.end = (uint64_t)j; m.bwt_start = (uint64_t)Jk; m.size = (int64_t)(uint64_t)Js; // e == j at every emit
        // (the extent of the read is looked up -- or, in an arena, reserved -- only by a fifth MEM: the first PGX_FAST_SLOTS have their own line)
        const uint64_t slot = nm < PGX_FAST_SLOTS ? 0ull : pgx_slot_extent(slot_off, slot_base, ovf_base, ovf_cap, n_ext_total, rid, nm, len, x, min_len);
        slots[pgx_slot_index(rid - chunk_first, chunk_reads, slot, nm)] = m;
        nm++;
        k = 0; kp = 0; s = n;
        // (as selects: an if / else that stores 1 into one of two flags is turned into ONE store through a selected address,
        //  which puts both flags into scratch memory)
        const bool more = j > x; // otherwise the loop of :722 runs zero times and the function returns j + 1
        ph = more ? 3 : ph;
        fresh = more ? 1u : fresh;
        x = more ? x : x + 1;
        restart = more ? restart : 1u;
    };

    for (;;) {
        // up until the probes' loads are out (pgx_dense2_pair lowers it again): see pgx_find_mems_pairs_kernel.  A launch that serves a list of reads (the
        // reads with a byte outside A C G T, on the second stream next to the pairs kernel: few, each a long chain) stays up: it is one wave per SIMD among
        // the other kernel's five, and at equal terms it took 5 to 19 ms from run to run -- longer than the pairs kernel it is meant to hide behind
        if (DENSE >= 2) __builtin_amdgcn_s_setprio(3);
        // ---- refill idle lanes ----
        unsigned long long idle = __ballot(ph == 0);
        while (idle) {
            if (rnext == rend) {
                if (exhausted) break;
                unsigned long long got = 0;
                if (lane == 0) got = first_read + atomicAdd(cursor, (unsigned long long)PGX_FM_BATCH); // the cursor counts from 0
                // wave-uniform values are moved to scalar registers explicitly
                got = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(got >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)got);
                if (got >= n_reads) { exhausted = true; break; }
                rnext = got;
                rend = got + PGX_FM_BATCH < n_reads ? got + PGX_FM_BATCH : n_reads;
            }
            const uint64_t avail = rend - rnext;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (ph == 0 && (uint64_t)rank < avail) {
                x = 0; nm = 0;
                rid = rnext + rank;
                if (rid_list) { const pgx_heavy_item it = rid_list[rid]; rid = it.rid; x = (int32_t)it.x; nm = it.nm; }
                base = offsets[rid];
                len = (int32_t)(offsets[rid + 1] - base);
                next0 = next;
                begin(); // may leave the lane idle again (read shorter than min_len)
                if (ph == 0) ph = -1; // served in this round; becomes idle again below
            }
            const uint32_t want = (uint32_t)__popcll(idle);
            rnext += (uint64_t)want < avail ? (uint64_t)want : avail;
            idle = __ballot(ph == 0);
        }
        if (ph == -1) ph = 0;
        if (!__any(ph > 0)) {
            if (exhausted && rnext == rend) break; // nothing live, nothing left
            continue;                               // only zero-work reads were handed out: refill again
        }
        // ---- one block decode for every live lane: an extension whose second probe falls outside the block
        //      of the first takes two trips of this loop (pend = 1 in between), so no lane ever waits for
        //      another lane's second trip ----
#ifdef PGX_FM_STATS
        st_trips++;
        st_live += (unsigned long long)__popcll(__ballot(ph > 0));
#endif
        // what the wave asks of the memory system in this trip (wave-uniform sums in scalar registers; images in global memory): seed / end table
        // entries -- one per first trip of a backward stage, an upper bound: windows that hold a byte outside A C G T and stages with fewer than
        // K extensions to go read the shared entry 0 -- and, counted behind the block, the lines holding the blocks of the two probes
        if (SEED) ln_seed += (unsigned long long)__popcll(__ballot(ph > 0 && fresh != 0u)); // (the seed table is in global memory whether or not the image is staged in LDS)
        bool c_blk = false, c_blk2 = false;
        if (ph > 0) {
            // ---- k-mer seed of a backward stage that starts now: the entry is loaded next to the block loads of the ordinary
            //      extension by P[j] (which every lane performs regardless) and replaces its result further down ----
            bool seed_lane = false;
            uint32_t kuse = 0u; // extensions the seed entry stands for
            uint4 se = make_uint4(0u, 0u, 0u, 0u);
            if (SEED) {
                const uint4 *sp = img.seed;
                if (fresh) {
                    // a stage that starts at j = len (step 3 of a MEM that reaches the end of its read) extends by 0 first, pattern[len]:
                    // the end table holds that extension followed by the seed_end_k bytes before the end of the read
                    const bool endw = j >= len;
                    const int32_t K = endw ? (int32_t)img.seed_end_k : (int32_t)img.seed_k;
                    const int32_t avail = (ph == 1) ? (j - x + 1) : (j - x); // extensions this stage may still perform
                    if (K && avail >= K + (endw ? 1 : 0)) {
                        const uint64_t a = base + (uint64_t)((endw ? len - 1 : j) - K + 1);
                        const uint32_t sh = (uint32_t)(a & 7ull) * 8u;
                        const uint64_t *wp = reinterpret_cast<const uint64_t *>(reads + (a & ~7ull)); // 32 zero bytes follow the last read
                        const uint64_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                        const uint64_t lo = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0, hi = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;
                        uint32_t sidx;
                        if (pgx_seed_index(lo, hi, (uint32_t)K, sidx)) { seed_lane = true; sp = (endw ? img.seed_end : img.seed) + sidx; kuse = (uint32_t)K + (endw ? 1u : 0u); }
                    }
                }
                fresh = 0u;
                se = *sp; // lanes without a seed read entry 0 (one cached line for all of them)
            }
            uint32_t byte = 0u; // pattern[len] reads as 0 (quirk 4)
            if (j < len) {
                const uint64_t at = base + (uint64_t)j;
                if (LDS_IMAGE) {
                    if ((at & ~7ull) != win_at) { // reads are padded with 16 zero bytes: the window never overruns
                        win_at = at & ~7ull;
                        win = *reinterpret_cast<const uint64_t *>(reads + win_at);
                    }
                    byte = (uint32_t)(win >> (8u * (uint32_t)(at & 7ull))) & 0xFFu;
                } else {
                    if ((at & ~31ull) != win_at) {
                        win_at = at & ~31ull;
                        const ulonglong2 w2 = *reinterpret_cast<const ulonglong2 *>(reads + win_at);
                        const ulonglong2 w3 = *reinterpret_cast<const ulonglong2 *>(reads + win_at + 16);
                        win = w2.x; win_hi = w2.y; win2 = w3.x; win3 = w3.y;
                    }
                    const uint64_t wlo = (at & 8ull) ? win_hi : win, whi = (at & 8ull) ? win3 : win2;
                    byte = (uint32_t)(((at & 16ull) ? whi : wlo) >> (8u * (uint32_t)(at & 7ull))) & 0xFFu;
                }
            }
            const bool fwd = (ph == 2);
            // extension by `byte` (backward, or forward = backward on the swapped interval by the complement,
            // folded into ext_tab[256 + byte]): src/r-index.cpp:713-764
            const uint32_t ee = s_ext[(fwd ? 256u : 0u) + byte];
            const uint32_t cv = PGX_EXT_CV(ee), mrow = PGX_EXT_M(ee);
            const pos_t kk = fwd ? kp : k, kq = fwd ? k : kp;
            bool fin;
            pos_t A1, dB;
            if (DENSE == 3) {
                // wide dense2: the same probes with 64-bit positions and superblock bases from LDS
                const uint64_t p0 = kk > n ? n : kk, p1 = (kk + s) > n ? n : (kk + s);
                uint64_t q0, q1, dq;
                pgx_dense2w_pair(img, s_sb, p0, p1, cv, mrow, q0, q1, dq, !rid_list);
                A0 = (pos_t)q0; A1 = (pos_t)q1; dB = (pos_t)dq;
                fin = true;
                c_blk = s != n;
                c_blk2 = c_blk && (uint32_t)(((p0 >> 7) * 0xAAAAAAABull) >> 33) != (uint32_t)(((p1 >> 7) * 0xAAAAAAABull) >> 33);
            } else if (DENSE == 2) {
                // dense2: header + one sub-block per probe, all within one 128-byte line (usually the same line for both probes)
                uint64_t q0, q1, dq;
                if (NARROW) {
                    const uint32_t ks = (uint32_t)kk + (uint32_t)s;
                    ovf |= ks < (uint32_t)kk;
                    const uint32_t p0 = (uint32_t)kk > (uint32_t)n ? (uint32_t)n : (uint32_t)kk, p1 = ks > (uint32_t)n ? (uint32_t)n : ks;
                    pgx_dense2_pair<true>(img, p0, p1, cv, mrow, q0, q1, dq, !rid_list);
                } else {
                    const uint64_t p0 = kk > n ? n : kk, p1 = (kk + s) > n ? n : (kk + s); // (kk + s wraps only from junk coordinates: either way >= n or tiny)
                    pgx_dense2_pair<false>(img, (uint32_t)p0, (uint32_t)p1, cv, mrow, q0, q1, dq, !rid_list);
                }
                A0 = (pos_t)q0; A1 = (pos_t)q1; dB = (pos_t)dq;
                fin = true;
                if (s != n) { // (the full interval probes block 0 and the last block: lines every lane shares)
                    const uint64_t e0 = (uint64_t)kk > (uint64_t)n ? (uint64_t)n : (uint64_t)kk, e1 = (uint64_t)kk + (uint64_t)s > (uint64_t)n ? (uint64_t)n : (uint64_t)kk + (uint64_t)s;
                    c_blk = true; c_blk2 = (uint32_t)((e0 * 0xAAAAAAABull) >> 40) != (uint32_t)((e1 * 0xAAAAAAABull) >> 40);
                }
            } else if (NARROW) {
                const uint32_t ks = (uint32_t)kk + (uint32_t)s;
                ovf |= ks < (uint32_t)kk; // kk + s left 32 bits (junk coordinates of a COMPAT quirk): the host repeats the chunk in 64 bits
                const uint32_t p0 = (uint32_t)kk > (uint32_t)n ? (uint32_t)n : (uint32_t)kk, p1 = ks > (uint32_t)n ? (uint32_t)n : ks;
                const PgxDenseBlk k0 = pgx_dense_load<LDS_IMAGE>(img, lds_blocks, p0), k1 = pgx_dense_load<LDS_IMAGE>(img, lds_blocks, p1);
                uint32_t a0, a1, d;
                pgx_dense_pair32(k0, p0, k1, p1, cv, mrow, a0, a1, d);
                A0 = (pos_t)a0; A1 = (pos_t)a1; dB = (pos_t)d;
                fin = true;
                c_blk = s != n; c_blk2 = c_blk && (p0 >> 7) != (p1 >> 7);
            } else if (DENSE == 1) {
                // dense image: the two block addresses are known at once (pos >> 6), so both 64-byte loads are in flight
                // together and every extension is a single trip
                const uint64_t p0 = kk > n ? n : kk, p1 = (kk + s) > n ? n : (kk + s);
                const PgxDenseBlk k0 = pgx_dense_load<LDS_IMAGE>(img, lds_blocks, p0), k1 = pgx_dense_load<LDS_IMAGE>(img, lds_blocks, p1);
                uint64_t Aq0, Aq1, Bq0, Bq1;
                pgx_dense_rank(k0, p0, cv, mrow, Aq0, Bq0);
                pgx_dense_rank(k1, p1, cv, mrow, Aq1, Bq1);
                A0 = (pos_t)Aq0; A1 = (pos_t)Aq1; dB = (pos_t)(Bq1 - Bq0);
                fin = true;
                c_blk = s != n; c_blk2 = c_blk && (p0 >> 7) != (p1 >> 7);
            } else if (LDS_IMAGE) {
                // image in LDS: no memory latency to hide and most extensions of a tiny index need both blocks,
                // so both trips run back to back (measured 6 % faster than the one-trip-per-iteration form)
                uint64_t Aq0, Aq1, dq;
                pgx_rank_pair<LDS_IMAGE, false>(img, lds_blocks, lds_dir, lds_blow, kk, kk + s, cv, mrow, Aq0, Aq1, dq);
                A0 = (pos_t)Aq0; A1 = (pos_t)Aq1; dB = (pos_t)dq;
                fin = true;
            } else {
                const uint64_t p0 = kk > n ? n : kk, p1 = (kk + s) > n ? n : (kk + s);
                uint64_t Ap, Bp, As, Bs;
                bool covered;
                pgx_probe<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_blow, pend ? p1 : p0, p1, !pend, cv, mrow, Ap, Bp, As, Bs, covered);
                c_blk = c_blk2 = s != n; // a directory entry and a 64-byte block per trip
                if (!pend) {
                    A0 = (pos_t)Ap; B0 = (pos_t)Bp;
                    A1 = (pos_t)As; dB = (pos_t)(Bs - Bp);
                    fin = covered;
                    pend = !covered;
                } else {
                    A1 = (pos_t)Ap; dB = (pos_t)(Bp - B0);
                    fin = true;
                    pend = false;
                }
            }
            if (fin) {
                next++;
                if (PGX_EXT_KILL(ee) || A0 >= A1) { // rank_k >= rank_ks -> bi_interval(0,0,0), src/r-index.cpp:751
                    k = 0; kp = 0; s = 0;
                } else {
                    const pos_t nk = A0 + (pos_t)s_C[PGX_EXT_V(ee)], nq = kq + dB;
                    if (NARROW) ovf |= nq < kq; // the other coordinate left 32 bits (see above)
                    s = A1 - A0;
                    k = fwd ? nq : nk;
                    kp = fwd ? nk : nq;
                }
                bool small = ((uint64_t)s < min_occ) || (s == 0); // :671 (unsigned compare) || size <= 0
                if (SEED && seed_lane) {
                    const uint32_t depth = se.w >> 24;
                    const pos_t ss = NARROW ? (pos_t)se.z : (pos_t)((uint64_t)se.z | ((uint64_t)((se.w >> 16) & 0xFFu) << 32));
                    if (ss != 0 && (uint64_t)ss >= min_occ) {
                        // all K extensions at once: sizes only shrink along a stage, so none of the K - 1 skipped ones was "small"
                        k = NARROW ? (pos_t)se.x : (pos_t)((uint64_t)se.x | ((uint64_t)(se.w & 0xFFu) << 32));
                        kp = NARROW ? (pos_t)se.y : (pos_t)((uint64_t)se.y | ((uint64_t)((se.w >> 8) & 0xFFu) << 32));
                        s = ss;
                        small = false;
                        j -= (int32_t)kuse - 1;
                        next += kuse - 1u;
                    } else if (ss == 0 && depth != PGX_SEED_UNUSABLE && min_occ <= 1) {
                        // the window leaves the index at its depth-th extension (only "empty" is small when min_occ <= 1)
                        k = 0; kp = 0; s = 0;
                        small = true;
                        j -= (int32_t)depth - 1;
                        next += depth - 1u;
                    } // otherwise (entry unusable, or min_occ decides where the stage ends): the ordinary extension stands
                }
                // The transitions of the three steps as selects (the 64 lanes of a wave are in all three steps at once, so
                // branches would run every path on every trip anyway, each with its own copies and exec-mask juggling):
                //   step 1  small -> restart at j + 1 | j == x -> J = interval, j = x + min_len, step 2 (or emit) | else j--
                //   step 2  small -> emit [x, j)      | else J = interval, j++, emit when j reaches len
                //   step 3  small -> restart at j + 1 | else j--, restart at x + 1 once j reaches x
                const bool adv = !small, p1 = ph == 1, p2 = ph == 2, at_x = j == x;
                const bool to2 = p1 && adv && at_x;
                const bool keep = adv && (to2 || p2);
                Jk = keep ? k : Jk;
                Js = keep ? s : Js;
                const int32_t jn = adv ? (p1 ? (at_x ? x + (int32_t)min_len : j - 1) : (p2 ? j + 1 : j - 1)) : j;
                const bool em = (p2 && (small || jn >= len)) || (to2 && jn >= len);
                const bool rs_small = small && !p2, rs_end = !p1 && !p2 && adv && jn <= x;
                restart = (rs_small || rs_end) ? 1u : 0u;
                x = rs_small ? j + 1 : (rs_end ? x + 1 : x);
                ph = to2 ? 2 : ph;
                j = jn;
                if (em) emit();       // x is unchanged in every emitting case
                if (restart) begin(); // next start position of this read (or the read is finished / handed on)
            }
        }
        if (!LDS_IMAGE) ln_blk += (unsigned long long)(__popcll(__ballot(c_blk)) + __popcll(__ballot(c_blk2)));
    }
    if (NARROW && __any(ovf) && lane == 0) n_ext_total[PGX_CTR_OVF32] = 1;
    // one atomic per wave for the extension counter
    unsigned long long tot = next;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    if (lane == 0 && tot) atomicAdd(n_ext_total + PGX_CTR_EXT, tot);
    if (lane == 0 && (ln_blk | ln_seed)) { atomicAdd(n_ext_total + PGX_CTR_FM_LINES, ln_blk); atomicAdd(n_ext_total + PGX_CTR_FM_SEEDS, ln_seed); }
#ifdef PGX_FM_STATS // wave trips, live lane-trips, longest wave (stats runs are made without tags)
    if (lane == 0) { atomicAdd(n_ext_total + PGX_CTR_ST_TRIPS, st_trips); atomicAdd(n_ext_total + PGX_CTR_ST_LIVE, st_live); atomicMax(n_ext_total + PGX_CTR_ST_LONGEST, st_trips); }
#endif
}

#define PGX_FM_INSTANTIATE(...)                                                                                                               \
    template __global__ void pgx_find_mems_kernel<__VA_ARGS__>(PgxDevImage, const uint8_t *, const uint64_t *, uint64_t, uint64_t, uint64_t,     \
                                                               const uint64_t *, pgx_mem *, uint32_t *, unsigned long long *, unsigned long long *, \
                                                               uint64_t, uint64_t, uint32_t, uint32_t, pgx_heavy_item *, unsigned long long *,     \
                                                               const pgx_heavy_item *, const unsigned long long *, uint32_t *, uint64_t);
PGX_FM_INSTANTIATE(false, 0, false, false)
PGX_FM_INSTANTIATE(false, 1, false, false)
PGX_FM_INSTANTIATE(true, 0, false, false)
PGX_FM_INSTANTIATE(true, 1, false, false)
PGX_FM_INSTANTIATE(true, 1, true, false)
PGX_FM_INSTANTIATE(true, 1, false, true)
PGX_FM_INSTANTIATE(true, 1, true, true)
PGX_FM_INSTANTIATE(false, 1, true, false)
PGX_FM_INSTANTIATE(false, 1, false, true)
PGX_FM_INSTANTIATE(false, 1, true, true)
PGX_FM_INSTANTIATE(false, 2, false, false)
PGX_FM_INSTANTIATE(false, 2, true, false)
PGX_FM_INSTANTIATE(false, 2, false, true)
PGX_FM_INSTANTIATE(false, 2, true, true)
PGX_FM_INSTANTIATE(false, 3, false, false)
PGX_FM_INSTANTIATE(false, 3, false, true)

// what the reads asked of the arena, for the host to size the next one: PGX_CTR_OVF_TOP = PGX_ARENA_SUBS x the fullest sub-arena's demand
__global__ void pgx_arena_demand_kernel(unsigned long long *__restrict__ ctr) {
    unsigned long long m = ctr[PGX_CTR_ARENA0 + 16u * threadIdx.x]; // (launched with PGX_ARENA_SUBS = 64 threads: one wave)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_down(m, off, 64); m = o > m ? o : m; }
    if (threadIdx.x == 0) ctr[PGX_CTR_OVF_TOP] = m * PGX_ARENA_SUBS;
}

// ------------------------------------------------------------------------------------------
// Reads with a byte outside A C G T (upper case): no seed applies to a window that holds one, so the two-step kernel could only hand
// them on after two trips -- and they are the long chains of the hand-on launch (a read cut from an N run has thousands of extensions).
// Found once per upload, they go to the dense2 kernel on a second stream WHILE the two-step kernel runs, which skips them.
// Two passes: a streaming one over the read bytes (16 per lane, coalesced) that lists the 16-byte chunks holding such a byte, and one
// thread per listed chunk that finds the reads its bad bytes belong to (binary search in the offsets), flags them and lists each once.
// (packed != NULL: the same pass writes the reads as two bits per symbol, 16 symbols per dword, A C T G = 0 1 2 3 -- the code order of the seed
//  index --, for pgx_find_mems_pairs_kernel<.., PACKED>; symbols of a chunk that holds another byte are junk, and so is what the reads flagged here stand for)
__global__ void __launch_bounds__(256)
pgx_bad_chunks_kernel(const uint8_t *__restrict__ reads, uint64_t n_bytes, uint64_t *__restrict__ chunks, unsigned long long *__restrict__ count, uint64_t cap,
                      uint32_t *__restrict__ packed) {
    const uint64_t n_chunks = (n_bytes + 15) >> 4; // (32 zero bytes follow the last read: the last chunk may be read whole)
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(reads + (c << 4));
        uint64_t b0, b1;
        const uint32_t c0 = pgx_seed_codes(v.x, b0);
        const uint32_t c1 = pgx_seed_codes(v.y, b1);
        if (packed) packed[c] = c0 | (c1 << 16);
        const uint64_t left = n_bytes - (c << 4); // bytes of the chunk that belong to reads
        if (left < 8) { b0 &= (1ull << (8 * left)) - 1ull; b1 = 0; }
        else if (left < 16) b1 &= (1ull << (8 * (left - 8))) - 1ull;
        if (b0 | b1) {
            const unsigned long long at = atomicAdd(count, 1ull);
            if (at < cap) chunks[at] = c;
        }
    }
}
__global__ void __launch_bounds__(256)
pgx_classify_reads_kernel(const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets, uint64_t n_reads, const uint64_t *__restrict__ chunks,
                          const unsigned long long *__restrict__ n_chunks, uint64_t cap, uint32_t *__restrict__ flag_words, pgx_heavy_item *__restrict__ list,
                          unsigned long long *__restrict__ count) {
    const uint64_t nc = *n_chunks < cap ? *n_chunks : cap;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const uint64_t p0 = chunks[i] << 4, total = offsets[n_reads];
    uint64_t rid = ~0ull, rend = 0;
    for (uint32_t k = 0; k < 16 && p0 + k < total; k++) {
        const uint64_t p = p0 + k;
        uint64_t bad;
        (void)pgx_seed_codes((uint64_t)reads[p] | 0x4141414141414100ull, bad); // the other seven bytes read as 'A'
        if (!(bad & 0xFFull)) continue;
        if (rid == ~0ull || p >= rend) { // the read holding byte p: the last one whose offset is <= p (empty reads hold nothing)
            uint64_t lo = 0, hi = n_reads;
            while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (offsets[mid + 1] <= p) lo = mid + 1; else hi = mid; }
            rid = lo; rend = offsets[rid + 1];
        }
        const uint32_t bit = 1u << (8u * (uint32_t)(rid & 3));
        if (!(atomicOr(flag_words + (rid >> 2), bit) & bit)) {
            pgx_heavy_item it;
            it.rid = rid; it.x = 0; it.nm = 0;
            list[atomicAdd(count, 1ull)] = it;
        }
    }
}

// Reads that arrive packed (pgx_batch_upload_packed: two bits per symbol from the host, a quarter of the bytes over the link): back to bytes for the
// kernels that read bytes ("ACTG"[code]; one packed word = 16 symbols per thread, one 16-byte store) ...
__global__ void __launch_bounds__(256)
pgx_unpack_reads_kernel(const uint32_t *__restrict__ packed, uint64_t n_chunks, uint8_t *__restrict__ reads) {
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = packed[c];
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v |= ((0x47544341u >> (8u * ((w >> (2 * (4 * q + k))) & 3u))) & 0xFFu) << (8 * k);
            o[q] = v;
        }
        *reinterpret_cast<uint4 *>(reads + (c << 4)) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
// ... and the reads that hold a byte outside A C G T, which the host lists with their bytes as they are: copied over what the packed words gave,
// flagged for the two-step kernel to skip and listed for the kernel that serves them (what pgx_bad_chunks_kernel + pgx_classify_reads_kernel
// find on the device when the reads arrive as bytes).  One 64-lane wave per listed read.
__global__ void __launch_bounds__(256)
pgx_side_reads_kernel(uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ side_ids, const uint64_t *__restrict__ side_off,
                      const uint8_t *__restrict__ side_bytes, uint64_t n_side, uint8_t *__restrict__ flags, pgx_heavy_item *__restrict__ list,
                      unsigned long long *__restrict__ count) {
    const uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (k >= n_side) return;
    const uint64_t rid = side_ids[k], dst = offsets[rid], len = offsets[rid + 1] - dst, src = side_off[k];
    for (uint64_t i = lane; i < len; i += 64) reads[dst + i] = side_bytes[src + i];
    if (lane == 0) {
        flags[rid] = 1;
        pgx_heavy_item it;
        it.rid = rid; it.x = 0; it.nm = 0;
        list[k] = it;
        if (k == 0) *count = n_side;
    }
}

// find_all_mems over the PAIRS image (pgx_image.h): the loop of pgx_find_mems_kernel, but a trip reads ONE 128-byte block that
// answers both ends of an interval (p1 within the block of p0; otherwise the interval runs on into the next block, which takes a
// second trip) and, where the stage has two more symbols to go, performs BOTH extensions from it.  With (c1, c2) the pair at a
// position, a the first symbol extended by and b the second:
//   first:   s1 = #{c1 = a} in [p0, p1),          k1 = C[a] + #{c1 = a} before p0,                      k' += #{c1 > a} in [p0, p1)
//   second:  s2 = #{c1 = a, c2 = b} in [p0, p1),  k2 = C[b] + #b before k1 + #{c1 = a, c2 = b} before p0,  k' += #{c1 = a, c2 > b} in [p0, p1)
//            (#b before k1 = pair_t2[a][b] + pairs (a, b) before p0: LF maps the positions with c1 = a onto [k1, k1 + s1), and BWT there is c2)
// ("> a": the regular symbols that sort after a, which is what the extension tables of such an index weight; counts that involve \n or N
// are zero in the ranges the kernel accepts.)  The first extension's result decides as in the stepwise search: if it is "small" the
// stage ends there and the second is dropped; otherwise the pair counts as two extensions, and the transitions below see the second
// one at its own j.  MEMs, restart positions and n_extensions are those of pgx_find_mems_kernel.  Positions, counts and C are below
// 2^32 (the image exists for such indexes only), so the state is 32-bit.  A stage that starts from the full interval takes its first
// extension from img.first_ext (or the seed tables): the image is never probed with the full interval.
// A lane that meets a flagged block or an interval wider than two blocks takes THAT extension through the image the PAIRS image accompanies
// (one rank probe after the other in a rolled loop, exact for every symbol) and carries on with pairs; until the end of round 3 it gave its read
// up to a list that pgx_find_mems_kernel served behind this kernel.
__device__ __forceinline__ uint32_t pgx_window_byte(uint64_t w0, uint64_t w1, uint64_t a) { // byte a of the 16-byte window
    return (uint32_t)(((a & 8ull) ? w1 : w0) >> (8u * (uint32_t)(a & 7ull))) & 0xFFu;
}
// WIDE: the 64-bit form (pgx_image.h "WIDE"): header counts are deltas against the bases of the block's superblock (staged in LDS), interval state,
// C and pair_t2 in 64 bits; everything else is the same kernel.
// PACKED: the reads as two bits per symbol (pgx_pack_reads_kernel: A C T G = 0 1 2 3, the order of the seed index), every lane's read copied
// into LDS when the lane takes it: the loop then reads its symbols (and whole seed windows, which ARE the seed index) from LDS instead of
// re-fetching 16-byte windows of the read bytes through L2 -- a fifth of the kernel's memory requests at chr22 scale (941 M requests per step of
// which 213 M were such windows: the ~300 k live reads do not stay in L2).  Only for launches that skip every read with a byte outside A C G T.
// COOP (needs PACKED): the wave fetches the 64 block lines of its lanes TOGETHER -- eight load instructions in which lanes 8 q .. 8 q + 7 read the
// eight 16-byte pieces of probe q's line (perfectly coalesced), through LDS -- instead of five loads per lane that each touch 64 different lines.
// For images beyond the reach of the address-translation caches (~3 GB: profiles/r03_ubench_gather_loads_per_line.txt) a random line costs one
// translation per load INSTRUCTION that touches it: 1 x 16 B of a line runs at 48 G lines/s, 5 x 16 B at 16-18 G/s, which is where the five-load
// probe sat on the 5.8 GB image of the 4.35e9-symbol index (17 G lines/s).
// S64: the image with a block every 64 positions (pgx_image.h): block b covers [64 b, 64 b + 96), so an interval of up to 32 positions
// never needs a second block; the second block of one that does overlaps the first by 32 positions and is read from position 32 on.
// LCE (needs PACKED; narrow images without COOP): the forward stage of a MEM over an interval of s <= img.lce_max occurrences is finished from the suffix array
// and the text (pgx_image.h "LCE image") instead of two symbols per line.  A trip compares what is left of the read with the text behind ONE occurrence of the
// interval (SA[k + i], three 16-byte loads from one or two lines of the 2-bit text) and then reads up to sixteen entries of img.lce_lcp, the common prefixes of
// neighbouring suffixes: occurrence t matches min(match of t - 1, lcp[k + t] - symbols matched before the stage), so an entry above the best match is one more
// occurrence of the final interval, one below it ends the stage (the matches of sorted suffixes with one pattern rise, stay, fall), one equal to it (or unknown)
// has occurrence t compared itself in the next trip.  The longest match and the occurrences that reach it -- consecutive in suffix order -- ARE what the stepwise
// extension would end with: MEM end = j + longest match, bwt_start = k + index of the first of them, size = their number; the extensions count as if made one by
// one (the failing one included).  ~1.6 trips per stage at 8 haplotypes instead of ~33.  min_occ <= 1 only (the longest match decides); a window that touches a
// line with an N or an endmarker sends the lane back to the stepwise path for that stage.  Without the table (PGX_FM_LCP=0) every occurrence is compared and only
// intervals of up to sixteen go this way.  In this variant a stage's first step (seed / first_ext entry) is applied at the top of the trip after the one in which
// the stage started (FUSE below).  Results are bit-identical (tests run all three ways).
template <bool SEED, bool WIDE, bool PACKED, bool COOP, bool S64, bool LCE>
#ifndef PGX_LCE_WAVES
#define PGX_LCE_WAVES PGX_FM_WAVES_PER_SIMD // (the text path holds its three pieces of text across the trip's body: 110 VGPRs; bounded to 96 it spills 56 bytes per lane: scripts/r4_exp7.sh)
#endif
__global__ void __launch_bounds__(PGX_FM_THREADS, LCE ? PGX_LCE_WAVES : ((PACKED && !WIDE && !COOP) ? PGX_PAIRS_PACKED_WAVES : PGX_FM_WAVES_PER_SIMD)) // (<= 96 VGPRs: five waves per SIMD fit and are what the launch uses; four are as fast -- 20.7 against 20.6-21.0 ms at chr22 scale --, three 21.8)
pgx_find_mems_pairs_kernel(PgxDevImage img, const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets,
                           uint64_t n_reads, uint64_t min_len, uint64_t min_occ, const uint64_t *__restrict__ slot_off,
                           pgx_mem *__restrict__ slots, uint32_t *__restrict__ mem_count, unsigned long long *__restrict__ n_ext_total,
                           unsigned long long *__restrict__ cursor, uint64_t first_read, uint64_t slot_base, uint32_t heavy_ext, uint32_t heavy_cap,
                           pgx_heavy_item *__restrict__ heavy_list, unsigned long long *__restrict__ heavy_count, const uint8_t *__restrict__ skip,
                           const uint32_t *__restrict__ packed, uint32_t pk_words, uint32_t *__restrict__ ovf_base, uint64_t ovf_cap) {
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type pos_t;
    __shared__ uint32_t s_ext[512];
    __shared__ pos_t s_C[8];
    __shared__ pos_t s_t2[32];
    // first_ext: [byte] the full interval extended by byte, [256 + byte] extended by 0 and then by byte (packed like a seed entry); PACKED (every byte is
    // one of A C G T): [code] by "ACTG"[code], [4] by 0, [5 + code] by 0 and then by "ACTG"[code]
    __shared__ uint4 s_fe[PACKED ? 16 : 512];
    extern __shared__ __align__(16) unsigned char pgx_dyn_lds[];
    static_assert(!COOP || PACKED, "the cooperative loads come with the packed reads");
    static_assert(!LCE || (PACKED && !WIDE && !COOP), "the text comparison reads the packed reads and 32-bit suffix array entries");
    constexpr uint32_t SYMS = PGX_PAIRS_SYMS, STRIDE = S64 ? PGX_PAIRS_STRIDE64 : PGX_PAIRS_SYMS;
    uint32_t *s_rd = reinterpret_cast<uint32_t *>(pgx_dyn_lds); // PACKED: word w of this thread's read at s_rd[w * blockDim.x + threadIdx.x] (pk_words words per thread)
    const uint32_t rd_stride = blockDim.x;
    // COOP: behind the packed reads, 8 KiB per wave: piece p of the line of lane q's probe at [q * 8 + (p ^ (q & 7))] (the swizzle spreads the banks)
    uint4 *s_stage = reinterpret_cast<uint4 *>(pgx_dyn_lds + (size_t)pk_words * PGX_FM_THREADS * 4) + (size_t)(threadIdx.x >> 6) * 512;
    // WIDE: behind those, per superblock the sixteen pair-count bases and their four row sums (24 words each, img.n_sbp superblocks)
    uint64_t *s_pb = reinterpret_cast<uint64_t *>(pgx_dyn_lds + (size_t)pk_words * PGX_FM_THREADS * 4 + (COOP ? (size_t)(PGX_FM_THREADS / 64) * 8192 : 0));
    // LCE (never with COOP / WIDE): behind the packed reads, what a lane has asked for at the end of a trip and uses in the next, fetched straight into LDS
    // (global_load_lds: no registers in between, nothing the compiler could copy too early): one 16-byte slot per lane -- the seed entry of a stage that starts --,
    // one dword -- the suffix array entry of the occurrence it compares with the text next --, five dwords -- sixteen entries of img.lce_lcp from any byte on
    // (dword t of lane l of wave w at s_lcp[(5 w + t) * 64 + l])
    uint4 *s_sa4 = reinterpret_cast<uint4 *>(pgx_dyn_lds + (size_t)pk_words * PGX_FM_THREADS * 4);
    uint32_t *s_sae = reinterpret_cast<uint32_t *>(s_sa4 + PGX_FM_THREADS);
    uint32_t *s_lcp = s_sae + PGX_FM_THREADS;
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) s_ext[i] = img.consts->ext_tab[i];
    if (threadIdx.x < 8) s_C[threadIdx.x] = (pos_t)img.consts->C[threadIdx.x];
    if (threadIdx.x < 32) s_t2[threadIdx.x] = (pos_t)img.consts->pair_t2w[threadIdx.x];
    if (WIDE) for (uint32_t i = threadIdx.x; i < img.n_sbp * 24u; i += blockDim.x) s_pb[i] = img.pbase[i];
    if (PACKED) { if (threadIdx.x < 9) s_fe[threadIdx.x] = img.first_ext[threadIdx.x == 4 ? 0u : (threadIdx.x > 4 ? 256u : 0u) + ((0x47544341u >> (8u * ((threadIdx.x > 4 ? threadIdx.x - 5u : threadIdx.x) & 3u))) & 0xFFu)]; }
    else for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) s_fe[i] = img.first_ext[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const pos_t n = (pos_t)img.n;
    const pos_t mo = WIDE ? (pos_t)min_occ : (pos_t)(min_occ > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)min_occ); // narrow: sizes are below 2^32, a larger min_occ makes everything "small" either way
    const bool mo_huge = !WIDE && min_occ > 0xFFFFFFFFull;
    // an entry of the seed tables / first_ext: {k lo, k' lo, s lo, k hi | k' hi << 8 | s hi << 16 | depth << 24}
    auto ent_k = [](const uint4 &e) { return WIDE ? (pos_t)((uint64_t)e.x | ((uint64_t)(e.w & 0xFFu) << 32)) : (pos_t)e.x; };
    auto ent_q = [](const uint4 &e) { return WIDE ? (pos_t)((uint64_t)e.y | ((uint64_t)((e.w >> 8) & 0xFFu) << 32)) : (pos_t)e.y; };
    auto ent_s = [](const uint4 &e) { return WIDE ? (pos_t)((uint64_t)e.z | ((uint64_t)((e.w >> 16) & 0xFFu) << 32)) : (pos_t)e.z; };
    uint32_t rid = 0; // (the launch serves fewer than 2^32 reads: pgx_batch_run)
    uint64_t base = 0;
    int32_t len = 0, x = 0, j = 0;
    pos_t k = 0, kp = 0, s = 0, Jk = 0, Js = 0;
    uint32_t nm = 0, next = 0, next0 = 0;
    int ph = 0;
    uint64_t win = 0, win_hi = 0;
    uint32_t win_at = ~0u; // the cached 16 bytes of the reads buffer: their offset / 16 (16 rather than 32 bytes: four registers less, no difference in time)
    uint32_t X0 = 0; // the four sums (each <= 96: one byte) over the first block of an interval that runs on into the next
    pos_t X0e = 0, X0f = 0;                                   // ... and the two absolute ranks at its start
    uint32_t pend = 0, fresh = 0, restart = 0;
    uint64_t rnext = 0, rend = 0;
    bool exhausted = false;
    unsigned long long ln_blk = 0, ln_seed = 0, ln_two = 0; // wave-uniform (scalar registers): block lines / seed entries the wave asked for, trips with two extensions (PGX_CTR_PAIRS_*)
    uint32_t did2 = 0; // this lane's last trip performed two extensions (summed at the top of the next trip, where the wave is converged)
    // LCE: bit 0 = the lane's stage goes through the text (ph == 2), bit 1 = this stage must not (a flagged text line), bit 2 = this trip only reads on in the
    // table of common prefixes (no occurrence is compared), bits 8..15 the occurrence / entry the trip starts with, 16..23 index of the first occurrence with
    // the longest match, 24..31 how many reach it; the longest match; the text position of the occurrence compared
    uint32_t lce_st = 0, lce_best = 0, lce_pos = 0;
    // FUSE: a stage's first trip (no line of the image: the seed / first_ext entry, then the transitions) is not a trip of its own.  The entry is asked for
    // at the END of the trip in which the stage starts (`fresh` 1 -> 2 | extensions the entry stands for << 8, the entry into se_pre) and applied at the top
    // of the next one, after which the lane takes part in that trip like any other: 6.7 of a 150-symbol read's 25 lane trips were such first trips.
    constexpr bool FUSE = LCE;
    // (the entry travels through LDS -- global_load_lds into the first of the lane's five suffix array pieces, which no stage that starts is using --: kept in
    //  registers, the compiler loaded it into others than the ones it lives in across the loop's back edge and copied it over there, behind a wait for the
    //  load, which put the entry's latency back into every trip)
    typedef uint32_t pgx_u32x4 __attribute__((ext_vector_type(4)));
#ifndef PGX_LCE_ENTRY_CAP
#define PGX_LCE_ENTRY_CAP 3u // with the common-prefix table a stage through the text is ~2 trips however wide the interval: worth it from 2 x 3 symbols to go
#endif
#ifndef PGX_FUSE2
#define PGX_FUSE2 0 // (1: two first steps per trip -- scripts/r4_exp13.sh: 11 % fewer wave trips, the same time: what a trip saves in number it costs in instructions)
#endif
#ifndef PGX_SEED_VIA_LDS
#define PGX_SEED_VIA_LDS 1 // (0: the entry in registers, loaded by the compiler -- scripts/r4_exp10.sh)
#endif
#ifndef PGX_LCE_ASM_LOADS
#define PGX_LCE_ASM_LOADS 1 // (0: the lines of the LCE variant loaded by the compiler)
#endif
    uint4 se_reg = make_uint4(0u, 0u, 0u, 0u);
    // (and so are the lines of the LCE variant: its two kinds of lanes load into the same registers at different points of a trip, and between them the
    //  compiler used those registers as scratch for the other kind -- after waiting for the first kind's loads, one memory latency in front of the other)
    pgx_u32x4 row = {0u, 0u, 0u, 0u}, hs = row, d0 = row, d1 = row, d2 = row; // (whoever reads them in a trip has loaded them in that trip)
    uint32_t lce_f0 = 0u, lce_f1 = 0u; // the flag words of the lines of a lane's text window
    auto ld4 = [](const uint4 *q) __attribute__((always_inline)) { return *reinterpret_cast<const pgx_u32x4 *>(q); };
#ifdef PGX_FM_STATS
    unsigned long long st_trips = 0, st_live = 0, st_wait = 0, st_fresh = 0; // diagnostics build only (scripts/fm_stats.sh)
    unsigned long long st_t_refill = 0, st_refills = 0, st_t_seed = 0, st_t_line = 0;
    const unsigned long long st_t0 = __builtin_readcyclecounter();
#endif

    auto begin = [&]() __attribute__((always_inline)) {
        if (x >= len || (uint64_t)(len - x) < min_len) { ph = 0; mem_count[rid] = nm; return; }
        if (heavy_ext && next - next0 >= heavy_ext && len <= (int32_t)PGX_FM_HEAVY_MAXLEN) {
            const unsigned long long at = atomicAdd(heavy_count, 1ull);
            if (at < (unsigned long long)heavy_cap) {
                pgx_heavy_item it;
                it.rid = (uint64_t)rid; it.x = (uint32_t)x; it.nm = nm;
                heavy_list[at] = it;
                ph = 0;
                return;
            }
        }
        k = 0; kp = 0; s = n;
        if (LCE) lce_st = 0u;
        if (min_len == 0) { Jk = 0; Js = n; j = x; ph = 2; }
        else { j = x + (int32_t)min_len - 1; ph = 1; fresh = 1u; }
    };
    // a lane on the text path asks for what its next trip reads: the suffix array entry of occurrence t (when it compares that one) and sixteen entries of the table
    // of common prefixes -- those of the occurrences behind t, or from entry t on
    auto lce_ask = [&](uint32_t k32, uint32_t t, bool cmp) __attribute__((always_inline)) {
        if (cmp) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(img.lce_sa + k32 + t), (void __attribute__((address_space(3))) *)(s_sae + (threadIdx.x >> 6) * 64u), 4, 0, 0);
        if (img.lce_lcp) {
            const uint32_t e0 = k32 + t + (cmp ? 1u : 0u), lb = e0 & ~3u;
#pragma unroll
            for (uint32_t q = 0; q < 5u; q++)
                if (q < 4u || (e0 & 3u) != 0u) // (sixteen bytes from e0 on: four dwords, five when e0 is not aligned)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(img.lce_lcp + lb + 4u * q),
                                                     (void __attribute__((address_space(3))) *)(s_lcp + ((threadIdx.x >> 6) * 5u + q) * 64u), 4, 0, 0);
        }
    };
    auto emit = [&]() __attribute__((always_inline)) {
        pgx_mem m;
        m.start = (uint64_t)x; m.end = (uint64_t)j; m.bwt_start = (uint64_t)Jk; m.size = (int64_t)(uint64_t)Js;
        // (the worst-case offset of the read is looked up only by a fifth MEM: the first PGX_FAST_SLOTS have their own line)
        const uint64_t slot = nm < PGX_FAST_SLOTS ? 0ull : pgx_slot_extent(slot_off, slot_base, ovf_base, ovf_cap, n_ext_total, (uint64_t)rid, nm, len, x, min_len);
        slots[pgx_slot_index((uint64_t)rid - first_read, n_reads - first_read, slot, nm)] = m;
        nm++;
        k = 0; kp = 0; s = n;
        if (LCE) lce_st = 0u;
        const bool more = j > x;
        ph = more ? 3 : ph;
        fresh = more ? 1u : fresh;
        x = more ? x : x + 1;
        restart = more ? restart : 1u;
    };

    for (;;) {
        // Wave priority: up from here until the trip's loads are out, down for the arithmetic on what they return.  The waves of a SIMD take turns
        // issuing; with equal priority they drift into step -- all computing, then all waiting -- and the memory pipeline idles in between.  A wave
        // that is about to ask for its lines now overtakes the ones that are counting bits: 17.3 -> 16.0-16.5 ms at chr22 scale
        // (profiles/r03_wave_priority.txt; the same priority for every wave, or the opposite order, is slower).
        __builtin_amdgcn_s_setprio(3);
        unsigned long long idle = __ballot(ph == 0);
        // (LCE: a read is ~28 lane trips now instead of ~73, so two or three lanes of a wave finish one in EVERY trip and a refill round -- two memory
        //  latencies in front of the trip's own -- ran in nine trips out of ten, a quarter of the waves' time: idle lanes now wait until img.refill_min of
        //  them have gathered, or until no lane of the wave is live)
        if (LCE && idle && (uint32_t)__popcll(idle) < img.refill_min && __popcll(idle) != 64) idle = 0ull;
#ifdef PGX_FM_STATS
        const unsigned long long st_r0 = __builtin_readcyclecounter();
        if (idle) st_refills++;
#endif
        while (idle) {
            if (rnext == rend) {
                if (exhausted) break;
                unsigned long long got = 0;
                if (lane == 0) got = first_read + atomicAdd(cursor, 32ull);
                got = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(got >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)got);
                if (got >= n_reads) { exhausted = true; break; }
                rnext = got;
                rend = got + 32ull < n_reads ? got + 32ull : n_reads;
            }
            const uint64_t avail = rend - rnext;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (ph == 0 && (uint64_t)rank < avail) {
                rid = (uint32_t)(rnext + rank);
                // (the three loads go out together, and so do the packed words below: a refill round is two memory latencies, not one per word --
                //  word by word, 58 % of the waves' time at chr22 scale went by in this loop: profiles/r03_refill_share.txt)
                const uint8_t *skp = skip ? skip + rid : reinterpret_cast<const uint8_t *>(offsets); // (always a load, never a branch with a wait of its own)
                const uint32_t skv = (uint32_t)*skp;
                const uint64_t o0 = offsets[rid], o1 = offsets[rid + 1];
                const uint32_t sk = skip ? skv : 0u;
                base = o0; // (assigned on both paths, so that the offsets are not fetched behind the branch on sk)
                len = (int32_t)(o1 - o0);
                if (sk) ph = -1; // served by the dense2 kernel on the other stream (pgx_classify_reads_kernel)
                else {
                    if (PACKED) { // the read's packed words into this thread's LDS column (the host sized pk_words for the longest read of the launch)
                        const uint32_t *src = packed + (base >> 4);
                        const uint32_t nw = ((uint32_t)(base & 15ull) + (uint32_t)len + 15u) >> 4;
#pragma unroll 1
                        for (uint32_t w0 = 0; w0 < nw; w0 += PGX_PK_GROUP) { // (a read of 150 symbols: ten or eleven words, one group)
                            uint32_t t[PGX_PK_GROUP];
#pragma unroll
                            for (uint32_t i = 0; i < PGX_PK_GROUP; i++) t[i] = src[w0 + i < nw ? w0 + i : nw - 1u];
#pragma unroll
                            for (uint32_t i = 0; i < PGX_PK_GROUP; i++) if (w0 + i < nw) s_rd[(w0 + i) * rd_stride + threadIdx.x] = t[i];
                        }
                    }
                    x = 0; nm = 0;
                    next0 = next;
                    begin();
                    if (ph == 0) ph = -1;
                }
            }
            const uint32_t want = (uint32_t)__popcll(idle);
            rnext += (uint64_t)want < avail ? (uint64_t)want : avail;
            idle = __ballot(ph == 0);
        }
#ifdef PGX_FM_STATS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        st_t_refill += __builtin_readcyclecounter() - st_r0;
#endif
        if (ph == -1) ph = 0;
        if (!__any(ph > 0)) {
            if (exhausted && rnext == rend) break;
            continue;
        }
#ifdef PGX_FM_STATS
        st_trips++;
        st_live += (unsigned long long)__popcll(__ballot(ph > 0));
        st_fresh += (unsigned long long)__popcll(__ballot(ph > 0 && fresh != 0u));
#endif
        // what the wave asks of the memory system in this trip (wave-uniform sums in scalar registers): a live lane fetches one block line, except in a
        // stage's first trip, which reads block 0 like every other such lane and takes its result from first_ext / the seed table
        // (seed / end table entries: one per first trip -- an upper bound: a stage with fewer than K extensions to go reads the shared entry 0)
        if (!FUSE) ln_blk += (unsigned long long)__popcll(__ballot(ph > 0 && fresh == 0u));
        ln_two += (unsigned long long)__popcll(__ballot(did2 != 0u));
        did2 = 0u;
        if (SEED && !FUSE) ln_seed += (unsigned long long)__popcll(__ballot(ph > 0 && fresh != 0u));
        if (COOP) { // every lane names the block it is about to probe (idle lanes: block 0, like the first trip of a stage), the wave fetches all 64 lines
            const pos_t kk_c = (ph == 2) ? kp : k;
            const uint32_t myblk = ph > 0 ? (S64 ? (uint32_t)(kk_c >> 6) : (uint32_t)(((uint64_t)(kk_c >> 5) * 0xAAAAAAABull) >> 33)) + pend : 0u;
            // (global_load_lds_dwordx4: straight into LDS, no registers for the data; lane l of instruction i lands at [64 i + l] = slot l & 7 of
            //  probe q = 8 i + (l >> 3), so the swizzle is applied to the piece it fetches)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t q = 8u * (uint32_t)i + ((uint32_t)lane >> 3), piece = ((uint32_t)lane & 7u) ^ (q & 7u);
                const uint32_t blk = (uint32_t)__shfl((int)myblk, (int)q, 64);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(img.pairs + ((size_t)blk * 8 + piece)),
                                                 (void __attribute__((address_space(3))) *)(s_stage + 64 * i), 16, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        // the five 16-byte pieces a lane loads in a trip: its block's row, counts and planes -- or, for a lane that compares with the text, three pieces of the
        // text, the flag words of the lines they lie in and the next occurrence's suffix array entry (the same registers: nothing added to the trip's pressure)
        if (LCE && PGX_LCE_ASM_LOADS) asm volatile("" : "=v"(row), "=v"(hs), "=v"(d0), "=v"(lce_f0), "=v"(lce_f1)); // (nothing of the last trip's lines is needed: the registers are free until here)
        const bool lce_lane = LCE && ph == 2 && (lce_st & 1u) != 0u;
        uint32_t lce_g0 = 0u;
        bool em_now = false; // this trip ends with a MEM (set by either kind of lane); `restart`: with the next start position of the read
        restart = 0u;
        if (LCE && __any(lce_lane)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (what the lanes on the text path asked for at the end of the last trip is in LDS)
#ifdef PGX_FM_STATS
        if (FUSE && __any(ph > 0 && fresh >= 2u)) {
            const unsigned long long st_s0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_t_seed += __builtin_readcyclecounter() - st_s0;
        }
#endif
        if (FUSE && PGX_SEED_VIA_LDS && __any(ph > 0 && fresh >= 0x100u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the entries asked for at the end of the last trip are in LDS)
        // the first extension(s) of a stage whose seed entry has arrived: what the stage's first trip does in the other variants
        auto prestep = [&]() __attribute__((always_inline)) {
            const uint32_t kuse = fresh >> 8; // extensions the seed entry stands for (0: none was asked for)
            const uint4 se_pre = PGX_SEED_VIA_LDS ? s_sa4[threadIdx.x] : se_reg;
            fresh = 0u;
            const bool at_end = j >= len, q1 = ph == 1;
            const uint32_t qa = (uint32_t)(base & 15ull) + (uint32_t)(at_end ? len : j);
            const uint32_t qs1 = at_end ? qa - 1u : qa, qs2 = qa ? qa - 1u : 0u;
            const uint32_t wa = s_rd[(qs1 >> 4) * rd_stride + threadIdx.x], wb = s_rd[(qs2 >> 4) * rd_stride + threadIdx.x];
            const uint4 f1 = s_fe[at_end ? 4u : ((wa >> (2u * (qs1 & 15u))) & 3u)], f2 = s_fe[5u + ((wb >> (2u * (qs2 & 15u))) & 3u)];
            const bool small1 = f1.z == 0u || f1.z < mo || mo_huge;
            const uint32_t sdepth = se_pre.w >> 24, se_s = se_pre.z;
            const bool seed_alive = kuse != 0u && se_s != 0u && se_s >= mo && !mo_huge;
            const bool seed_dead = kuse != 0u && se_s == 0u && sdepth != PGX_SEED_UNUSABLE && min_occ <= 1;
            const bool rem2 = q1 ? (j - 1 >= x) : (j - 1 > x);
            const bool do2 = at_end && rem2 && !seed_alive && !seed_dead && !small1; // (by 0, then by the last symbol of the read: quirk 4)
            uint32_t ns = do2 ? f2.z : f1.z, nk = do2 ? f2.x : f1.x, nq = do2 ? f2.y : f1.y;
            if (ns == 0u) { nk = 0u; nq = 0u; }
            j -= do2 ? 1 : 0;
            next += do2 ? 2u : 1u;
            did2 = do2 ? 1u : 0u;
            s = ns; k = nk; kp = nq;
            bool small = ns == 0u || ns < mo || mo_huge;
            if (seed_alive) {
                k = se_pre.x; kp = se_pre.y; s = se_s;
                small = false;
                j -= (int32_t)kuse - 1;
                next += kuse - 1u;
            } else if (seed_dead) {
                k = 0u; kp = 0u; s = 0u;
                small = true;
                j -= (int32_t)sdepth - 1;
                next += sdepth - 1u;
            }
            const bool adv = !small, at_x = j == x;
            const bool to2 = q1 && adv && at_x;
            Jk = to2 ? k : Jk;
            Js = to2 ? s : Js;
            const int32_t jn = adv ? (q1 ? (at_x ? x + (int32_t)min_len : j - 1) : j - 1) : j;
            const bool rs_end = !q1 && adv && jn <= x;
            restart = (small || rs_end) ? 1u : 0u;
            x = small ? j + 1 : (rs_end ? x + 1 : x);
            ph = to2 ? 2 : ph;
            j = jn;
            em_now = to2 && jn >= len;
        };
        // the stages that have just started (in the refill round, by a restart or behind a MEM): their seed entries are asked for
        auto prefetch = [&]() __attribute__((always_inline)) {
            const bool ask = ph > 0 && fresh == 1u;
            bool asked = false;
            if (ask) {
                const bool endw = j >= len; // (the end table: see pgx_find_mems_kernel)
                const int32_t K = endw ? (int32_t)img.seed_end_k : (int32_t)img.seed_k;
                const int32_t avail = (ph == 1) ? (j - x + 1) : (j - x);
                uint32_t kuse = 0u;
                if (K && avail >= K + (endw ? 1 : 0)) { // the window's 2 K bits of the packed read are the index
                    const uint32_t q = (uint32_t)(base & 15ull) + (uint32_t)((endw ? len - 1 : j) - K + 1);
                    const uint32_t w0 = s_rd[(q >> 4) * rd_stride + threadIdx.x], w1 = s_rd[((q >> 4) + 1u) * rd_stride + threadIdx.x]; // (one word of padding per thread)
                    const uint32_t sidx = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (2u * (q & 15u))) & (uint32_t)((1ull << (2 * K)) - 1ull); // (K = 16: all 32 bits)
                    const uint4 *sp = (endw ? img.seed_end : img.seed) + sidx;
                    if (PGX_SEED_VIA_LDS) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)sp, (void __attribute__((address_space(3))) *)(s_sa4 + (threadIdx.x >> 6) * 64u), 16, 0, 0);
                    else se_reg = *sp;
                    kuse = (uint32_t)K + (endw ? 1u : 0u);
                    asked = true;
                }
                fresh = 2u | (kuse << 8);
            }
            ln_seed += (unsigned long long)__popcll(__ballot(asked));
        };
        if (FUSE && ph > 0 && fresh >= 2u) prestep();
        // (experiment, off: a stage that ENDS in its first step -- a dead seed entry, step 3 behind a seed: 6.8 of a read's 12 lane trips -- starts the next one at
        //  once and asks for ITS entry now: it arrives with this trip's lines and is applied behind them (second call below) -- two such steps per trip.)
        if (FUSE && PGX_FUSE2) {
            if (restart) { begin(); restart = 0u; }
            prefetch();
        }
        const bool sit_out = FUSE && (fresh != 0u || em_now || restart != 0u); // no line of the image for this lane in this trip
        if (FUSE) ln_blk += (unsigned long long)__popcll(__ballot(ph > 0 && !sit_out));
        const bool lce_cmp = lce_lane && (lce_st & 4u) == 0u; // this trip compares an occurrence with the text
        if (LCE && lce_cmp) {
            lce_pos = s_sae[threadIdx.x];
            lce_g0 = lce_pos + (uint32_t)(j - x);           // text position that faces read symbol j
            const uint32_t w0 = lce_g0 >> 4;                // its word (16 symbols); the window: words w0 .. w0 + 11
            const uint32_t *tp = img.lce_text + w0;
            // (dword-aligned 16-byte pieces; nothing waits for them here: section C does)
            if (PGX_LCE_ASM_LOADS)
                asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dwordx4 %2, %3, off offset:32"
                             : "+v"(row), "+v"(hs), "+v"(d0) : "v"(tp) : "memory");
            else {
                typedef pgx_u32x4 __attribute__((aligned(4))) u4a_t;
                row = *reinterpret_cast<const u4a_t *>(tp); hs = *reinterpret_cast<const u4a_t *>(tp + 4); d0 = *reinterpret_cast<const u4a_t *>(tp + 8);
            }
            const uint32_t l0 = w0 >> 5, l1 = (w0 + 11u) >> 5; // the lines of the window
            const uint32_t *fp0 = img.lce_flags + (l0 >> 5), *fp1 = img.lce_flags + (l1 >> 5);
            if (PGX_LCE_ASM_LOADS) asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off" : "+v"(lce_f0), "+v"(lce_f1) : "v"(fp0), "v"(fp1) : "memory");
            else { lce_f0 = *fp0; lce_f1 = *fp1; }
        }
        if (LCE) ln_blk += (unsigned long long)(__popcll(__ballot(lce_cmp && ((lce_g0 >> 4) >> 5) != (((lce_g0 >> 4) + 11u) >> 5))) + // a window over two lines
                                                __popcll(__ballot(lce_cmp)) +                                                    // the line of the suffix array entry
                                                (img.lce_lcp ? __popcll(__ballot(lce_lane && s > 1u)) : 0)) -                      // ... and of the common prefixes
                             (unsigned long long)__popcll(__ballot(lce_lane && !lce_cmp));                                        // (no text line in such a trip: counted above as one)
        if (ph > 0 && !lce_lane && !sit_out) {
            const bool fr = !FUSE && fresh != 0u; // first extension of a backward stage: from first_ext / the seed table
            bool seed_lane = false;
            uint32_t kuse = 0u; // extensions the seed entry stands for
            uint4 se = make_uint4(0u, 0u, 0u, 0u);
            if (SEED && !FUSE) {
                const uint4 *sp = img.seed;
                if (fr) {
                    const bool endw = j >= len; // (the end table: see pgx_find_mems_kernel)
                    const int32_t K = endw ? (int32_t)img.seed_end_k : (int32_t)img.seed_k;
                    const int32_t avail = (ph == 1) ? (j - x + 1) : (j - x);
                    if (K && avail >= K + (endw ? 1 : 0)) {
                        if (PACKED) { // the window's 2 K bits of the packed read are the index
                            const uint32_t q = (uint32_t)(base & 15ull) + (uint32_t)((endw ? len - 1 : j) - K + 1);
                            const uint32_t w0 = s_rd[(q >> 4) * rd_stride + threadIdx.x], w1 = s_rd[((q >> 4) + 1u) * rd_stride + threadIdx.x]; // (one word of padding per thread)
                            const uint32_t sidx = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (2u * (q & 15u))) & (uint32_t)((1ull << (2 * K)) - 1ull); // (K = 16: all 32 bits)
                            seed_lane = true; sp = (endw ? img.seed_end : img.seed) + sidx; kuse = (uint32_t)K + (endw ? 1u : 0u);
                        } else {
                        const uint64_t a = base + (uint64_t)((endw ? len - 1 : j) - K + 1);
                        const uint32_t sh = (uint32_t)(a & 7ull) * 8u;
                        const uint64_t *wp = reinterpret_cast<const uint64_t *>(reads + (a & ~7ull));
                        const uint64_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                        const uint64_t lo = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0, hi = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;
                        uint32_t sidx;
                        if (pgx_seed_index(lo, hi, (uint32_t)K, sidx)) { seed_lane = true; sp = (endw ? img.seed_end : img.seed) + sidx; kuse = (uint32_t)K + (endw ? 1u : 0u); }
                        }
                    }
                }
#ifdef PGX_FM_STATS
                const unsigned long long st_s0 = __builtin_readcyclecounter();
#endif
                se = *sp;
#ifdef PGX_FM_STATS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st_t_seed += __builtin_readcyclecounter() - st_s0;
#endif
            }
            if (!FUSE) fresh = 0u;
            const bool fwd = (ph == 2);
            const uint64_t at = base + (uint64_t)j;
            // pattern[len] reads as 0 (quirk 4): step 3 of a MEM that reaches the end of its read starts there, from the full interval;
            // its first TWO extensions (by 0, then by the last symbol of the read) come from first_ext: the rows of the endmarkers
            // (block 0 of the image) have the sequences' last symbols before them, often N
            const bool at_end = j >= len;
            uint32_t byte, byte2;
            bool have2;
            if (PACKED) { // both symbols from the packed read in LDS ("ACTG"[code]); the second one is always at hand
                const uint32_t qa = (uint32_t)(base & 15ull) + (uint32_t)(at_end ? len : j);
                const uint32_t q1 = at_end ? qa - 1u : qa, q2 = (fwd && !at_end) ? qa + 1u : (qa ? qa - 1u : 0u); // (q2 is meaningless where the stage has no second symbol: rem2)
                const uint32_t wa = s_rd[(q1 >> 4) * rd_stride + threadIdx.x], wb = s_rd[(q2 >> 4) * rd_stride + threadIdx.x];
                byte = at_end ? 0u : ((0x47544341u >> (8u * ((wa >> (2u * (q1 & 15u))) & 3u))) & 0xFFu);
                byte2 = (0x47544341u >> (8u * ((wb >> (2u * (q2 & 15u))) & 3u))) & 0xFFu;
                have2 = true;
            } else {
            const uint64_t atw = at_end ? at - 1ull : at; // (a live read has len >= 1)
            if ((uint32_t)(atw >> 4) != win_at) {
                win_at = (uint32_t)(atw >> 4);
                const ulonglong2 w2 = *reinterpret_cast<const ulonglong2 *>(reads + (atw & ~15ull));
                win = w2.x; win_hi = w2.y;
            }
            // (a function of values: as a lambda capturing the window by reference it turned into loads through a selected address,
            //  with the window in scratch memory)
            byte = at_end ? 0u : pgx_window_byte(win, win_hi, at);
            // the symbol after this one in the direction of the stage, when the cached window holds it
            const uint64_t at2 = (fwd && !at_end) ? at + 1ull : at - 1ull;
            have2 = (uint32_t)(at2 >> 4) == win_at;
            byte2 = pgx_window_byte(win, win_hi, at2);
            }
            const uint32_t e1 = s_ext[(fwd ? 256u : 0u) + byte], e2 = s_ext[(fwd ? 256u : 0u) + byte2];
            const uint32_t cv1 = PGX_EXT_CV(e1), cv2 = PGX_EXT_CV(e2);
            const bool reg1 = !PGX_EXT_KILL(e1) && ((0x2Eu >> cv1) & 1u), reg2 = !PGX_EXT_KILL(e2) && ((0x2Eu >> cv2) & 1u); // A C G T
            const uint32_t t1 = reg1 ? cv1 - 1u - (cv1 >> 2) : 0u, t2 = reg2 ? cv2 - 1u - (cv2 >> 2) : 0u;                    // their 2-bit codes
            const bool rem2 = ph == 1 ? (j - 1 >= x) : (fwd ? (j + 1 < len) : (j - 1 > x)); // the stage has a second extension to make
            const bool two = !fr && rem2 && have2 && reg1 && reg2;
            const pos_t kk = fwd ? kp : k, kq = fwd ? k : kp;
            const pos_t p0 = kk, p1 = kk + s;
            // the block of p0 (96 positions); a second trip (pend) reads the block after it
            const uint32_t bfirst = S64 ? (uint32_t)(p0 >> 6) : (uint32_t)(((uint64_t)(p0 >> 5) * 0xAAAAAAABull) >> 33); // p0 / 96 (p0 < 2^37)
            const pos_t endrel_p = p1 - (pos_t)bfirst * STRIDE;                               // p1 relative to the first block
            const uint32_t endrel = endrel_p > (pos_t)0xFFFFu ? 0xFFFFu : (uint32_t)endrel_p; // (anything beyond two blocks is "far")
            // the second block starts STRIDE positions after the first and the first has answered up to its position SYMS
            const uint32_t relA = pend ? SYMS - STRIDE : p0 - bfirst * STRIDE;
            const uint32_t relB = pend ? endrel - STRIDE : (endrel < SYMS ? endrel : SYMS);
            if (COOP) { // the line of this lane's probe is in LDS (fetched by the whole wave above)
                const uint4 *mine = s_stage + (uint32_t)lane * 8u;
                const uint32_t sw = (uint32_t)lane & 7u;
                row = ld4(mine + (t1 ^ sw)); hs = ld4(mine + (4u ^ sw)); d0 = ld4(mine + (5u ^ sw)); d1 = ld4(mine + (6u ^ sw)); d2 = ld4(mine + (7u ^ sw));
            } else {
#ifdef PGX_FM_STATS
                const unsigned long long st_l0 = __builtin_readcyclecounter();
#endif
                const uint4 *bp = img.pairs + (size_t)(bfirst + pend) * 8;
                // row: pairs (t1, A C G T) before the block; hs: positions before the block with c2 special and c1 = A, C, G, T; bit 31 of .x: flag;
                // d0 d1 d2: the planes: c1 bit 0, c1 bit 1, c2 bit 0, c2 bit 1, three dwords each
                if (LCE && PGX_LCE_ASM_LOADS) { // (in place, and waited for by hand below: see the declaration of row)
                    const uint4 *rp = bp + t1;
                    asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\tglobal_load_dwordx4 %2, %4, off offset:80"
                                 : "+v"(row), "+v"(hs), "+v"(d0) : "v"(rp), "v"(bp) : "memory");
                    d1 = ld4(bp + 6); d2 = ld4(bp + 7); // (only this kind of lane uses these two)
                } else {
                    row = ld4(bp + t1); hs = ld4(bp + 4); d0 = ld4(bp + 5); d1 = ld4(bp + 6); d2 = ld4(bp + 7);
                }
#ifdef PGX_EXP_LOAD6 // sensitivity experiment (scripts/exp_dup.sh): a sixth piece of the same line, thrown away
                { const uint4 xx = bp[(t1 + 1u) & 3u]; asm volatile("" ::"v"(xx.x), "v"(xx.y), "v"(xx.z), "v"(xx.w)); }
#endif
                __builtin_amdgcn_s_setprio(0);
#ifdef PGX_FM_STATS
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(row), "+v"(hs), "+v"(d0), "+v"(d1), "+v"(d2) :: "memory");
                st_t_line += __builtin_readcyclecounter() - st_l0;
#endif
            }
            // masks that turn "code == t" / "code > t" into plane expressions: (x ^ i0) & (y ^ i1) and (y & ua) | (x & (y | va) & wa)
            const uint32_t i0 = (t1 & 1u) ? 0u : 0xFFFFFFFFu, i1 = (t1 & 2u) ? 0u : 0xFFFFFFFFu, j0 = (t2 & 1u) ? 0u : 0xFFFFFFFFu, j1 = (t2 & 2u) ? 0u : 0xFFFFFFFFu;
            const uint32_t ua = t1 < 2u ? 0xFFFFFFFFu : 0u, va = t1 == 0u ? 0xFFFFFFFFu : 0u, wa = (t1 & 1u) ? 0u : 0xFFFFFFFFu;
            const uint32_t ub = t2 < 2u ? 0xFFFFFFFFu : 0u, vb = t2 == 0u ? 0xFFFFFFFFu : 0u, wb = (t2 & 1u) ? 0u : 0xFFFFFFFFu;
            if (LCE && !COOP && PGX_LCE_ASM_LOADS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(row), "+v"(hs), "+v"(d0) :: "memory"); // (behind what does not need the line)
            const bool flagged = (hs.x >> 31) != 0u;
            const uint32_t pts = (t1 == 0u ? hs.x : (t1 == 1u ? hs.y : (t1 == 2u ? hs.z : hs.w))) & 0x00FFFFFFu; // (24-bit counts: pgx_image.h)
            const uint32_t PX[3] = {d0.x, d0.y, d0.z}, PY[3] = {d0.w, d1.x, d1.y}, PU[3] = {d1.z, d1.w, d2.x}, PV[3] = {d2.y, d2.z, d2.w};
            // counts below relA (absolute ranks need them) and in [relA, relB) (sizes and the other coordinate are differences)
            uint32_t e1p = 0, e2p = 0, e1r = 0, g1r = 0, e2r = 0, g2r = 0;
#pragma unroll
            for (int h = 0; h < 3; h++) {
                const int32_t ta = (int32_t)relA - 32 * h, tb = (int32_t)relB - 32 * h;
                const uint32_t mP = ta >= 32 ? 0xFFFFFFFFu : (ta > 0 ? ((1u << ta) - 1u) : 0u);
                const uint32_t mR = (tb >= 32 ? 0xFFFFFFFFu : (tb > 0 ? ((1u << tb) - 1u) : 0u)) & ~mP;
                const uint32_t x = PX[h], y = PY[h], u = PU[h], v = PV[h];
                const uint32_t m1 = (x ^ i0) & (y ^ i1);             // first symbol == the one extended by
                const uint32_t g1 = (y & ua) | (x & (y | va) & wa);   // first symbol sorts after it
                const uint32_t q2 = m1 & (u ^ j0) & (v ^ j1);         // ... and second symbol == the second one extended by
                const uint32_t g2 = m1 & ((v & ub) | (u & (v | vb) & wb)); // ... and second symbol sorts after it
                e1p += __popc(m1 & mP); e2p += __popc(q2 & mP);
                e1r += __popc(m1 & mR); g1r += __popc(g1 & mR); e2r += __popc(q2 & mR); g2r += __popc(g2 & mR);
            }
#ifdef PGX_EXP_DUP // sensitivity experiment (scripts/exp_dup.sh): the popcount section once more on the same planes, results thrown away
            {
                uint32_t dsum = 0;
#pragma unroll
                for (int h = 0; h < 3; h++) {
                    uint32_t x = PX[h], y = PY[h], u = PU[h], v = PV[h];
                    asm volatile("" : "+v"(x), "+v"(y), "+v"(u), "+v"(v));
                    const int32_t ta = (int32_t)relA - 32 * h, tb = (int32_t)relB - 32 * h;
                    const uint32_t mP = ta >= 32 ? 0xFFFFFFFFu : (ta > 0 ? ((1u << ta) - 1u) : 0u);
                    const uint32_t mR = (tb >= 32 ? 0xFFFFFFFFu : (tb > 0 ? ((1u << tb) - 1u) : 0u)) & ~mP;
                    const uint32_t m1 = (x ^ i0) & (y ^ i1);
                    const uint32_t g1 = (y & ua) | (x & (y | va) & wa);
                    const uint32_t q2 = m1 & (u ^ j0) & (v ^ j1);
                    const uint32_t g2 = m1 & ((v & ub) | (u & (v | vb) & wb));
                    dsum += __popc(m1 & mP) + __popc(q2 & mP) + __popc(m1 & mR) + __popc(g1 & mR) + __popc(q2 & mR) + __popc(g2 & mR);
                }
                asm volatile("" ::"v"(dsum));
            }
#endif
            pos_t a01 = (pos_t)(row.x + row.y + row.z + row.w + pts + e1p);                       // rank of the first symbol at p0
            pos_t a02 = (pos_t)((t2 == 0u ? row.x : (t2 == 1u ? row.y : (t2 == 2u ? row.z : row.w))) + e2p); // rank of the pair at p0
            if (WIDE) { // the counts of a block are deltas against its superblock
                const uint64_t *pb = s_pb + (size_t)((bfirst + pend) >> img.pairs_sb_shift) * 24u;
                a01 += (pos_t)pb[16u + t1];
                a02 += (pos_t)pb[4u * t1 + t2];
            }
            const bool straddle = endrel > SYMS, far = endrel > STRIDE + SYMS;
            // Run continuation (pgx_image.h): the pair of the block's last position goes on for hs.y >> 24 positions behind the block, its first symbol
            // alone for hs.z >> 24.  An interval that ends inside that stretch is answered from THIS line -- both extensions, or the first one --: the
            // positions behind the block all count like the last one.  (Intervals are ~#haplotypes wide and mostly one run: with 96 haplotypes 44 % of
            // the lane trips fetched a second block before, profiles/r04_haps_sweep.txt.)
            const uint32_t over = endrel - SYMS; // (meaningful where straddle)
            const bool cont2 = !pend && straddle && over <= (hs.y >> 24), cont1 = !pend && straddle && !cont2 && over <= (hs.z >> 24);
            const bool cont = cont1 || cont2;
            if (cont) {
                const uint32_t l1 = (PX[2] >> 31) | ((PY[2] >> 31) << 1), l2 = (PU[2] >> 31) | ((PV[2] >> 31) << 1); // the pair at position 95
                e1r += l1 == t1 ? over : 0u;
                g1r += l1 > t1 ? over : 0u;
                e2r += (cont2 && l1 == t1 && l2 == t2) ? over : 0u;
                g2r += (cont2 && l1 == t1 && l2 > t2) ? over : 0u;
            }
            const bool bail = !fr && (flagged || (far && !cont)); // (a second block is used only when it is not flagged either: nothing special between the two ends)
            const bool wait = !fr && !pend && straddle && !bail && !cont; // the interval runs on into the next block: next trip
#ifdef PGX_FM_STATS
            st_wait += wait ? 1ull : 0ull;
#endif
            if (wait) {
                X0 = e1r | (g1r << 8) | (e2r << 16) | (g2r << 24); X0e = a01; X0f = a02;
                pend = 1u;
            } else {
                const uint32_t Xp = pend ? X0 : 0u;
                const uint32_t c1 = (Xp & 0xFFu) + e1r, w1 = ((Xp >> 8) & 0xFFu) + g1r, c2 = ((Xp >> 16) & 0xFFu) + e2r, w2 = (Xp >> 24) + g2r;
                const pos_t r1 = pend ? X0e : a01, r2 = pend ? X0f : a02;
                pend = 0u;
                // first extension (src/r-index.cpp:713-764); a symbol that is not A C G T has no occurrence in a range free of special positions
                pos_t s1 = reg1 ? (pos_t)c1 : (pos_t)0;
                pos_t k1 = r1 + s_C[PGX_EXT_V(e1)], q1v = kq + (pos_t)w1;
                if (bail) { // special positions in the way (or an interval wider than two blocks): THIS extension alone through the dense2 image the
                    // PAIRS image accompanies (two more lines, exact for every symbol), then on with pairs.  (Until round 3 the read went to a list
                    // and pgx_find_mems_kernel searched the rest of it behind this kernel: 0.8 ms of a 20 ms step for 0.1 % of the reads.)
                    // (one probe after the other in a rolled loop: this path is rare and must not cost the common one its registers)
                    const uint64_t q0 = (uint64_t)p0 > (uint64_t)n ? (uint64_t)n : (uint64_t)p0, q1 = (uint64_t)p1 > (uint64_t)n ? (uint64_t)n : (uint64_t)p1;
                    uint64_t A0 = 0, A1 = 0, B0 = 0, B1 = 0;
#pragma unroll 1
                    for (int it = 0; it < 2; it++) {
                        const uint64_t q = it ? q1 : q0;
                        uint64_t a, bq;
                        if (WIDE) pgx_dense2w_rank(img, q, PGX_EXT_CV(e1), PGX_EXT_M(e1), a, bq);
                        else if (img.dense == 1) { // (a small index: the 64-byte dense image)
                            const PgxDenseBlk db = pgx_dense_load<false>(img, nullptr, q);
                            pgx_dense_rank(db, q, PGX_EXT_CV(e1), PGX_EXT_M(e1), a, bq);
                        } else pgx_dense2_rank(img, (uint32_t)q, PGX_EXT_CV(e1), PGX_EXT_M(e1), a, bq);
                        A1 = a; B1 = bq;
                        if (!it) { A0 = a; B0 = bq; }
                    }
                    const uint64_t dB = B1 - B0;
                    { // (rare, and inside diverged control flow: counted with one atomic by the first lane that is here, not in the wave-uniform sums)
                        const unsigned long long here = __ballot(true);
                        if (lane == (int)__ffsll((long long)here) - 1) {
                            atomicAdd(n_ext_total + PGX_CTR_REDO, (unsigned long long)__popcll(here));
                            atomicAdd(n_ext_total + PGX_CTR_PAIRS_LINES, 2ull * (unsigned long long)__popcll(here)); // the two lines of the other image
                        }
                    }
                    const bool none = PGX_EXT_KILL(e1) || A0 >= A1; // rank_k >= rank_ks -> bi_interval(0,0,0), src/r-index.cpp:751
                    s1 = none ? (pos_t)0 : (pos_t)(A1 - A0);
                    k1 = (pos_t)A0 + s_C[PGX_EXT_V(e1)];
                    q1v = kq + (pos_t)dB;
                }
                if (fr) { const uint4 f = s_fe[PACKED ? (at_end ? 4u : ((byte >> 1) & 3u)) : byte]; k1 = ent_k(f); q1v = ent_q(f); s1 = ent_s(f); }
                const bool small1 = s1 == 0u || s1 < mo || mo_huge;
                // a usable seed entry stands for the first extension and the ones after it
                const uint32_t sdepth = se.w >> 24;
                const pos_t se_s = ent_s(se);
                const bool seed_alive = SEED && seed_lane && se_s != 0u && se_s >= mo && !mo_huge;
                const bool seed_dead = SEED && seed_lane && se_s == 0u && sdepth != PGX_SEED_UNUSABLE && min_occ <= 1;
                const bool do2 = (fr ? (at_end && rem2 && have2 && !seed_alive && !seed_dead) : (two && !bail && !cont1)) && !small1;
                pos_t s2 = (pos_t)c2, k2 = r2 + s_C[PGX_EXT_V(e2)] + s_t2[8u * t1 + cv2], q2v = q1v + (pos_t)w2;
                if (fr) { const uint4 f = s_fe[PACKED ? 5u + ((byte2 >> 1) & 3u) : 256u + byte2]; k2 = ent_k(f); q2v = ent_q(f); s2 = ent_s(f); }
                pos_t ns = do2 ? s2 : s1, nk = do2 ? k2 : k1, nq = do2 ? q2v : q1v;
                if (ns == 0u) { nk = 0u; nq = 0u; }
                if (do2) { // the first of the two: what a trip of its own would have left behind
                    Jk = fwd ? q1v : Jk;
                    Js = fwd ? s1 : Js;
                    j = fwd ? j + 1 : j - 1;
                }
                next += do2 ? 2u : 1u;
                did2 = do2 ? 1u : 0u;
                s = ns;
                k = fwd ? nq : nk;
                kp = fwd ? nk : nq;
                bool small = ns == 0u || ns < mo || mo_huge;
                if (seed_alive) { // all its extensions at once: sizes only shrink along a stage, so none of the skipped ones was "small"
                    k = ent_k(se); kp = ent_q(se); s = se_s;
                    small = false;
                    j -= (int32_t)kuse - 1;
                    next += kuse - 1u;
                } else if (seed_dead) { // the window leaves the index at its depth-th extension
                    k = 0u; kp = 0u; s = 0u;
                    small = true;
                    j -= (int32_t)sdepth - 1;
                    next += sdepth - 1u;
                }
                const bool adv = !small, q1 = ph == 1, q2 = ph == 2, at_x = j == x;
                const bool to2 = q1 && adv && at_x;
                const bool keep = adv && (to2 || q2);
                Jk = keep ? k : Jk;
                Js = keep ? s : Js;
                const int32_t jn = adv ? (q1 ? (at_x ? x + (int32_t)min_len : j - 1) : (q2 ? j + 1 : j - 1)) : j;
                const bool em = (q2 && (small || jn >= len)) || (to2 && jn >= len);
                const bool rs_small = small && !q2, rs_end = !q1 && !q2 && adv && jn <= x;
                restart = (rs_small || rs_end) ? 1u : 0u;
                x = rs_small ? j + 1 : (rs_end ? x + 1 : x);
                ph = to2 ? 2 : ph;
                j = jn;
                em_now = em;
                // a forward stage over a narrow interval goes on through the text: from the next trip on one occurrence per trip (its first suffix array entry
                // is asked for now); only where that is fewer trips than two symbols per trip, and where the window of three pieces holds what is left
                if (LCE && img.lce_sa && ph == 2 && !em && !restart && !(lce_st & 2u) && s >= 1u && (uint32_t)s <= img.lce_max && mo <= 1u && j < len &&
                    (uint32_t)(len - j) >= 2u * (img.lce_lcp && (uint32_t)s > PGX_LCE_ENTRY_CAP ? PGX_LCE_ENTRY_CAP : (uint32_t)s) && (uint32_t)(len - j) <= 144u &&
                    ((uint32_t)s <= 16u || (img.lce_lcp && (uint32_t)(len - x) <= PGX_LCP_CAP - 1u))) { // (wider than sixteen only where the table of common prefixes can be used)
                    lce_st = 1u;
                    lce_best = 0u;
                    lce_ask((uint32_t)k, 0u, true); // SA[k] and the common prefixes of the sixteen entries behind it
                }
            }
        }
        if (LCE && PGX_LCE_ASM_LOADS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(row), "+v"(hs), "+v"(d0), "+v"(lce_f0), "+v"(lce_f1) :: "memory"); // (the whole wave: the text of section A is in)
        if (FUSE && PGX_FUSE2 && !PGX_LCE_ASM_LOADS && __any(ph > 0 && fresh >= 0x100u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FUSE && PGX_FUSE2 && ph > 0 && fresh >= 2u) prestep(); // (the entries asked for in front of this trip's lines are in LDS: see above)
        if (LCE && lce_lane) { // the text behind occurrence i of the interval against the read from symbol j on, then sixteen entries of the table
            const uint32_t rem = (uint32_t)(len - j), i = (lce_st >> 8) & 0xFFu;
            uint32_t a = (lce_st >> 16) & 0xFFu, cnt = lce_st >> 24, t0 = i;
            bool fin = false, banned = false;
            if (lce_cmp) {
                const uint32_t T[12] = {row.x, row.y, row.z, row.w, hs.x, hs.y, hs.z, hs.w, d0.x, d0.y, d0.z, d0.w};
                const uint32_t fl0 = lce_g0 >> 9, fl1 = ((lce_g0 >> 4) + 11u) >> 5; // the lines of the window (section A)
                banned = (((lce_f0 >> (fl0 & 31u)) | (lce_f1 >> (fl1 & 31u))) & 1u) != 0u; // a line with an N / an endmarker / behind the text: this stage goes on stepwise (nothing has changed yet)
                const uint32_t q0 = (uint32_t)(base & 15ull) + (uint32_t)j;
                const uint32_t tsh = 2u * (lce_g0 & 15u), rsh = 2u * (q0 & 15u), rw0 = q0 >> 4;
                uint32_t R[10];
#pragma unroll
                for (uint32_t u = 0; u < 10; u++) { const uint32_t wi = rw0 + u; R[u] = s_rd[(wi < pk_words ? wi : pk_words - 1u) * rd_stride + threadIdx.x]; }
                uint32_t l = 144u;
#pragma unroll
                for (int u = 8; u >= 0; u--) { // (from the last unit down: the first differing one wins)
                    const uint32_t df = __builtin_amdgcn_alignbit(R[u + 1], R[u], rsh) ^ __builtin_amdgcn_alignbit(T[u + 1], T[u], tsh);
                    l = df ? 16u * (uint32_t)u + ((uint32_t)__builtin_ctz(df) >> 1) : l;
                }
                l = l < rem ? l : rem;
                // (the matches of one pattern with suffixes in sorted order rise, stay, fall and never rise again: one below the best ends the stage)
                const bool better = i == 0u || l > lce_best;
                fin = !better && l < lce_best;
                cnt = better ? 1u : (l == lce_best ? cnt + 1u : cnt);
                a = better ? i : a;
                lce_best = better ? l : lce_best;
                t0 = i + 1u;
            }
            if (banned) lce_st = 2u;
            else {
                // The occurrences behind, from the common prefixes of neighbouring suffixes: with c symbols shared between occurrence t - 1 and t, occurrence t matches
                // min(match of t - 1, c - m) symbols (m = the symbols matched before the stage).  Here the match of t - 1 is the best one (anything shorter has ended the
                // stage): an entry above it (or at it, once the read is used up) is one more occurrence of the final interval; one below it ends the stage; one AT it --
                // occurrence t may match further -- or an unknown one is compared with the text itself in the next trip.  Sixteen entries per trip.
                uint32_t tt = t0; // the entry the next trip starts with
                bool cmp_next = true;
                if (!fin && tt < (uint32_t)s && img.lce_lcp && (uint32_t)(len - x) <= PGX_LCP_CAP - 1u) { // (m + what is left of the read stays below the cap: a capped entry is "longer than anything asked")
                    const uint32_t m = (uint32_t)(j - x), bsh = ((uint32_t)k + t0) & 3u;
                    uint32_t L5[5];
#pragma unroll
                    for (uint32_t t = 0; t < 5u; t++) L5[t] = s_lcp[((threadIdx.x >> 6) * 5u + t) * 64u + (uint32_t)lane];
                    const uint32_t W[4] = {__builtin_amdgcn_alignbyte(L5[1], L5[0], bsh), __builtin_amdgcn_alignbyte(L5[2], L5[1], bsh), __builtin_amdgcn_alignbyte(L5[3], L5[2], bsh),
                                           __builtin_amdgcn_alignbyte(L5[4], L5[3], bsh)};
                    bool stop = false;
                    cmp_next = false;
#pragma unroll
                    for (uint32_t u = 0; u < 16u; u++) {
                        if ((u & 3u) == 0u && u > 0u && !__any(!stop && t0 + u < (uint32_t)s)) break; // (four at a time)
                        const uint32_t c = (W[u >> 2] >> (8u * (u & 3u))) & 0xFFu, rel = c - m;
                        const bool act = !stop && t0 + u < (uint32_t)s;
                        const bool hard = c == PGX_LCP_UNKNOWN || c < m || (rel == lce_best && lce_best < rem);
                        const bool drop = !hard && rel < lce_best;
                        cmp_next = cmp_next || (act && hard);
                        fin = fin || (act && drop);
                        stop = stop || (act && (hard || drop));
                        if (act && !hard && !drop) { cnt++; tt = t0 + u + 1u; }
                    }
                }
                if (!fin && tt < (uint32_t)s) { // on with occurrence / entry tt
                    lce_st = 1u | (cmp_next ? 0u : 4u) | (tt << 8) | (a << 16) | (cnt << 24);
                    lce_ask((uint32_t)k, tt, cmp_next);
                } else { // the MEM ends where the longest match ends; the occurrences that reach it are its interval
                    Jk = k + (pos_t)a; Js = (pos_t)cnt;
                    next += lce_best + (lce_best < rem ? 1u : 0u); // (the extension that fails counts, as in the stepwise stage)
                    j += (int32_t)lce_best;
                    lce_st = 0u;
                    em_now = true;
                }
            }
        }
        if (em_now) emit();
        if (restart) begin();
        if (FUSE) prefetch(); // (their seed entries are on the way while the wave loops)
    }
    unsigned long long tot = next;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    if (lane == 0 && tot) atomicAdd(n_ext_total + PGX_CTR_EXT, tot);
    ln_two += (unsigned long long)__popcll(__ballot(did2 != 0u));
    if (lane == 0 && ln_blk) { atomicAdd(n_ext_total + PGX_CTR_PAIRS_LINES, ln_blk); atomicAdd(n_ext_total + PGX_CTR_PAIRS_SEEDS, ln_seed); atomicAdd(n_ext_total + PGX_CTR_PAIRS_TWO, ln_two); }
#ifdef PGX_FM_STATS // wave trips / live lane-trips, lane-trips waiting for the second block / fresh
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) st_wait += __shfl_down(st_wait, off, 64);
    if (lane == 0) { atomicAdd(n_ext_total + PGX_CTR_ST_PAIR_T_REFILL, st_t_refill); atomicAdd(n_ext_total + PGX_CTR_ST_PAIR_T_TOTAL, __builtin_readcyclecounter() - st_t0); atomicAdd(n_ext_total + PGX_CTR_ST_PAIR_REFILLS, st_refills);
                     atomicAdd(n_ext_total + PGX_CTR_ST_PAIR_T_SEED, st_t_seed); atomicAdd(n_ext_total + PGX_CTR_ST_PAIR_T_LINE, st_t_line); }
    if (lane == 0) { atomicAdd(n_ext_total + PGX_CTR_ST_PAIR_TRIPS, st_trips); atomicAdd(n_ext_total + PGX_CTR_ST_PAIR_LIVE, st_live); atomicAdd(n_ext_total + PGX_CTR_ST_PAIR_WAIT, st_wait); atomicAdd(n_ext_total + PGX_CTR_ST_PAIR_FRESH, st_fresh); }
#endif
}
#define PGX_PAIRS_INSTANTIATE(...)                                                                                                                \
    template __global__ void pgx_find_mems_pairs_kernel<__VA_ARGS__>(PgxDevImage, const uint8_t *, const uint64_t *, uint64_t, uint64_t, uint64_t, const uint64_t *, \
                                                                     pgx_mem *, uint32_t *, unsigned long long *, unsigned long long *, uint64_t, uint64_t, uint32_t, \
                                                                     uint32_t, pgx_heavy_item *, unsigned long long *, const uint8_t *, const uint32_t *, uint32_t,  \
                                                                     uint32_t *, uint64_t);
PGX_PAIRS_INSTANTIATE(true, false, false, false, false, false)
PGX_PAIRS_INSTANTIATE(true, true, false, false, false, false)
PGX_PAIRS_INSTANTIATE(true, false, true, false, false, false)
PGX_PAIRS_INSTANTIATE(true, true, true, false, false, false)
PGX_PAIRS_INSTANTIATE(true, false, true, true, false, false)
PGX_PAIRS_INSTANTIATE(true, true, true, true, false, false)
PGX_PAIRS_INSTANTIATE(true, false, false, false, true, false)
PGX_PAIRS_INSTANTIATE(true, true, false, false, true, false)
PGX_PAIRS_INSTANTIATE(true, false, true, false, true, false)
PGX_PAIRS_INSTANTIATE(true, true, true, false, true, false)
PGX_PAIRS_INSTANTIATE(true, false, true, true, true, false)
PGX_PAIRS_INSTANTIATE(true, true, true, true, true, false)
PGX_PAIRS_INSTANTIATE(true, false, true, false, false, true)
PGX_PAIRS_INSTANTIATE(true, false, true, false, true, true)

// first extension of every backward stage: the full interval extended by each byte value
__global__ void __launch_bounds__(256) pgx_first_ext_kernel(PgxDevImage img, uint4 *__restrict__ out) { // out[512]
    __shared__ uint32_t s_ext[512];
    __shared__ uint64_t s_C[8];
    pgx_stage_tables<false>(img, s_ext, s_C, nullptr, nullptr, nullptr);
    uint64_t k = 0, kp = 0, s = img.n;
    pgx_extend<false>(img, nullptr, nullptr, nullptr, s_ext, s_C, k, kp, s, threadIdx.x, false);
    auto pack = [](uint64_t k_, uint64_t q_, uint64_t s_) { // like a seed entry: low dwords, then the bits 32..39 of each
        return make_uint4((uint32_t)k_, (uint32_t)q_, (uint32_t)s_, (uint32_t)((k_ >> 32) & 0xFFu) | ((uint32_t)((q_ >> 32) & 0xFFu) << 8) | ((uint32_t)((s_ >> 32) & 0xFFu) << 16));
    };
    out[threadIdx.x] = pack(k, kp, s);
    k = 0; kp = 0; s = img.n; // by 0 (what pattern[len] reads as), then by the byte
    pgx_extend<false>(img, nullptr, nullptr, nullptr, s_ext, s_C, k, kp, s, 0u, false);
    if (s) pgx_extend<false>(img, nullptr, nullptr, nullptr, s_ext, s_C, k, kp, s, threadIdx.x, false);
    out[256 + threadIdx.x] = pack(k, kp, s);
}

// find_mems_function(pattern, min_len, min_occ, x) (algorithm.hpp:653-736) for ONE start position: the MEM it emits (if any), the
// start position it returns and the extensions it performs.
template <bool LDS_IMAGE>
__device__ __forceinline__ PgxHeavyResult pgx_fmf_eval(const PgxDevImage &img, const uint4 *lds_blocks, const uint64_t *lds_dir, const uint16_t *lds_blow,
                                                       const uint32_t *s_ext, const uint64_t *s_C, const uint8_t *__restrict__ pat, int32_t len, int32_t xs,
                                                       uint64_t min_len, uint64_t min_occ) {
    const uint64_t n = img.n;
    PgxHeavyResult r;
    r.mem.start = (uint64_t)xs; r.mem.end = 0; r.mem.bwt_start = 0; r.mem.size = 0;
    r.next_x = (uint32_t)len; r.n_ext = 0; r.has_mem = 0; r.pad = 0;
    if ((uint64_t)(len - xs) >= min_len) {
        uint64_t k = 0, kp = 0, s = n;
        uint32_t ne = 0;
        bool dead = false;
        for (int32_t j = xs + (int32_t)min_len - 1; j >= xs; j--) { // step 1 (:666-676)
            pgx_extend<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_blow, s_ext, s_C, k, kp, s, pat[j], false);
            ne++;
            if (s < min_occ || s == 0) { r.next_x = (uint32_t)(j + 1); dead = true; break; }
        }
        if (!dead) {
            uint64_t Jk = k, Js = s;
            int32_t j = xs + (int32_t)min_len;
            for (; j < len; j++) { // step 2 (:684-696)
                pgx_extend<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_blow, s_ext, s_C, k, kp, s, pat[j], true);
                ne++;
                if (s < min_occ || s == 0) break;
                Jk = k; Js = s;
            }
            r.has_mem = 1;
            r.mem.end = (uint64_t)j; r.mem.bwt_start = Jk; r.mem.size = (int64_t)Js; // :713
            k = 0; kp = 0; s = n;
            uint32_t nxt = (uint32_t)(xs + 1);
            for (; j > xs; j--) { // step 3 (:718-735); pattern[len] reads 0
                pgx_extend<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_blow, s_ext, s_C, k, kp, s, j < len ? pat[j] : (uint8_t)0, false);
                ne++;
                if (s < min_occ || s == 0) { nxt = (uint32_t)(j + 1); break; }
            }
            r.next_x = nxt;
        }
        r.n_ext = ne;
    }
    return r;
}

// ------------------------------------------------------------------------------------------
// The rest of a heavy read (see pgx_find_mems_kernel): one workgroup per read evaluates find_mems_function(x)
// (algorithm.hpp:653-736) for EVERY remaining start position x at once -- the calls are independent of each other,
// only the choice of the next start is a chain -- and thread 0 then follows the chain through the stored results,
// writing the MEMs of the visited starts in order and adding only their extensions to the extension counter.  About
// twice the extensions of the sequential walk, ~300 of them on the critical path instead of ~len^2 / 2.
template <bool LDS_IMAGE>
__global__ void __launch_bounds__(256)
pgx_find_mems_heavy_kernel(PgxDevImage img, const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets, uint64_t min_len,
                           uint64_t min_occ, const uint64_t *__restrict__ slot_off, uint64_t slot_base, pgx_mem *__restrict__ slots,
                           uint32_t *__restrict__ mem_count, unsigned long long *__restrict__ n_ext_total,
                           const pgx_heavy_item *__restrict__ heavy_list, const unsigned long long *__restrict__ heavy_count,
                           uint32_t heavy_cap, PgxHeavyResult *__restrict__ scratch, uint64_t chunk_first, uint64_t chunk_reads,
                           uint32_t *__restrict__ ovf_base, uint64_t ovf_cap) {
    unsigned long long cnt = *heavy_count;
    if (cnt == 0) return; // the usual case: nothing was handed on (uniform exit before any staging)
    if (cnt > heavy_cap) cnt = heavy_cap;
    __shared__ uint32_t s_ext[512];
    __shared__ uint64_t s_C[8];
    PGX_LDS_CARVE(img);
    pgx_stage_tables<LDS_IMAGE>(img, s_ext, s_C, lds_blocks, lds_dir, lds_blow);
    PgxHeavyResult *res = scratch + (size_t)blockIdx.x * PGX_FM_HEAVY_MAXLEN;
    for (unsigned long long h = blockIdx.x; h < cnt; h += gridDim.x) {
        const pgx_heavy_item it = heavy_list[h];
        const uint64_t base = offsets[it.rid];
        const int32_t len = (int32_t)(offsets[it.rid + 1] - base), x0 = (int32_t)it.x;
        const uint8_t *pat = reads + base;
        for (int32_t xs = x0 + (int32_t)threadIdx.x; xs < len; xs += (int32_t)blockDim.x) {
            const PgxHeavyResult r = pgx_fmf_eval<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_blow, s_ext, s_C, pat, len, xs, min_len, min_occ);
            res[xs - x0] = r;
        }
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t nm = it.nm;
            unsigned long long ne = 0;
            int32_t x = x0;
            while (x < len && (uint64_t)(len - x) >= min_len) {
                const PgxHeavyResult r = res[x - x0];
                ne += r.n_ext;
                if (r.has_mem) {
                    const uint64_t slot = nm < PGX_FAST_SLOTS ? 0ull : pgx_slot_extent(slot_off, slot_base, ovf_base, ovf_cap, n_ext_total, it.rid, nm, len, x, min_len);
                    slots[pgx_slot_index(it.rid - chunk_first, chunk_reads, slot, nm)] = r.mem; nm++;
                }
                x = (int32_t)r.next_x;
            }
            mem_count[it.rid] = nm;
            atomicAdd(n_ext_total, ne);
        }
        __syncthreads(); // res is reused by the next item of this workgroup
    }
}
template __global__ void pgx_find_mems_heavy_kernel<false>(PgxDevImage, const uint8_t *, const uint64_t *, uint64_t, uint64_t, const uint64_t *, uint64_t,
                                                           pgx_mem *, uint32_t *, unsigned long long *, const pgx_heavy_item *,
                                                           const unsigned long long *, uint32_t, PgxHeavyResult *, uint64_t, uint64_t, uint32_t *, uint64_t);
template __global__ void pgx_find_mems_heavy_kernel<true>(PgxDevImage, const uint8_t *, const uint64_t *, uint64_t, uint64_t, const uint64_t *, uint64_t,
                                                          pgx_mem *, uint32_t *, unsigned long long *, const pgx_heavy_item *,
                                                          const unsigned long long *, uint32_t, PgxHeavyResult *, uint64_t, uint64_t, uint32_t *, uint64_t);

// ------------------------------------------------------------------------------------------
// primitives for tests (mirror rank_at_cached_encoded / backward_extend_encoded / forward_...)
__global__ void __launch_bounds__(256)
pgx_rank_kernel(PgxDevImage img, const uint64_t *__restrict__ pos, uint64_t n, int true_codes, uint64_t *__restrict__ out) {
    // one (position, slot) per thread
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 6 * n) return;
    const uint64_t i = t / 6;
    const uint32_t sl = (uint32_t)(t - 6 * i);
    const uint32_t sigma = img.consts->sigma;
    uint64_t A = 0, B;
    if (true_codes) pgx_rank_ab<false>(img, nullptr, nullptr, nullptr, pos[i], sl, 0, A, B);
    else if (sl < sigma) pgx_rank_ab<false>(img, nullptr, nullptr, nullptr, pos[i], img.consts->slot_code[sl], 0, A, B);
    out[t] = A;
}
// Probe of the round-3 anomaly (DESIGN.md "stale counts"; scripts/anomaly_probe.py; PGX_RANK_PROBE selects it in pgx_rank_batch): the shape
// pgx_rank_kernel had until commit 66308c2 -- LOOP: one thread walks the six slots of its position -- and the form of pos / 384 it used -- MULHI --,
// each switchable on its own, over a wide dense2 image with true codes.  Test-only.
template <bool LOOP, bool MULHI>
__global__ void __launch_bounds__(256)
pgx_rank_probe_kernel(PgxDevImage img, const uint64_t *__restrict__ pos, uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (LOOP) {
        if (t >= n) return;
        for (uint32_t sl = 0; sl < 6; sl++) {
            uint64_t A = 0, B;
            const uint64_t p = pos[t] > img.n ? img.n : pos[t];
            pgx_dense2w_rank<MULHI>(img, p, sl, 0, A, B);
            out[t * 6 + sl] = A;
        }
    } else {
        if (t >= 6 * n) return;
        const uint64_t i = t / 6;
        uint64_t A = 0, B;
        const uint64_t p = pos[i] > img.n ? img.n : pos[i];
        pgx_dense2w_rank<MULHI>(img, p, (uint32_t)(t - 6 * i), 0, A, B);
        out[t] = A;
    }
}
template __global__ void pgx_rank_probe_kernel<true, true>(PgxDevImage, const uint64_t *, uint64_t, uint64_t *);
template __global__ void pgx_rank_probe_kernel<true, false>(PgxDevImage, const uint64_t *, uint64_t, uint64_t *);
template __global__ void pgx_rank_probe_kernel<false, true>(PgxDevImage, const uint64_t *, uint64_t, uint64_t *);

template <bool LDS_IMAGE>
__global__ void __launch_bounds__(256)
pgx_extend_kernel(PgxDevImage img, const pgx_biint *__restrict__ in, const uint8_t *__restrict__ sym,
                  const uint8_t *__restrict__ forward, uint64_t n, pgx_biint *__restrict__ out) {
    __shared__ uint32_t s_ext[512];
    __shared__ uint64_t s_C[8];
    PGX_LDS_CARVE(img);
    pgx_stage_tables<LDS_IMAGE>(img, s_ext, s_C, lds_blocks, lds_dir, lds_blow);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = in[i].forward, kp = in[i].reverse, s = (uint64_t)in[i].size;
    pgx_extend<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_blow, s_ext, s_C, k, kp, s, sym[i], forward[i] != 0);
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
//   mode 3: in8[i] == (uint8_t)min_len  (indicator; merge_tags)        mode 4: in8[i]        mode 5: in64[i], 1 -> 0
__device__ __forceinline__ uint64_t pgx_scan_load(int mode, const void *in, uint64_t i, uint64_t min_len) {
    if (mode == 0) return ((const uint32_t *)in)[i];
    if (mode == 1) return ((const uint64_t *)in)[i];
    if (mode == 3) return ((const uint8_t *)in)[i] == (uint8_t)min_len ? 1u : 0u;
    if (mode == 4) return ((const uint8_t *)in)[i];
    if (mode == 5) { const uint64_t v = ((const uint64_t *)in)[i]; return v == 1 ? 0 : v; } // tag segments: single runs need none
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
pgx_scan_partial_kernel(int mode, const void *in, uint64_t n_cap, uint64_t min_len, uint64_t *__restrict__ block_sums, const uint64_t *__restrict__ n_dev) {
    __shared__ uint64_t s_wave[4];
    const uint64_t n = n_dev ? (*n_dev < n_cap ? *n_dev : n_cap) : n_cap; // the actual count may live on the device (speculative sizing)
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
pgx_scan_apply_kernel(int mode, const void *in, uint64_t n_cap, uint64_t min_len, const uint64_t *__restrict__ block_sums,
                      uint64_t nb, uint64_t *__restrict__ out, uint64_t *__restrict__ total_out, int raw_sums, const uint64_t *__restrict__ n_dev) {
    __shared__ uint64_t s_wave[4];
    const uint64_t n = n_dev ? (*n_dev < n_cap ? *n_dev : n_cap) : n_cap;
    const uint64_t b0 = (uint64_t)blockIdx.x * 256 * PGX_SCAN_ITEMS;
    // raw_sums: block_sums holds the per-block totals as pgx_scan_partial_kernel wrote them (few blocks: every block adds up
    // the totals before it, which saves the single-block launch in between); otherwise their exclusive scan + grand total
    uint64_t base, grand = 0;
    if (raw_sums) {
        uint64_t part = 0, all = 0;
        for (uint64_t i = threadIdx.x; i < nb; i += 256) {
            const uint64_t t = block_sums[i];
            part += i < blockIdx.x ? t : 0;
            all += t;
        }
        uint64_t tot;
        (void)pgx_block_excl_scan(part, s_wave, tot);
        base = tot;
        if (blockIdx.x == 0) { (void)pgx_block_excl_scan(all, s_wave, tot); grand = tot; }
    } else {
        base = block_sums[blockIdx.x];
        grand = block_sums[nb];
    }
    uint64_t vals[PGX_SCAN_ITEMS], v = 0;
    for (int t = 0; t < PGX_SCAN_ITEMS; t++) {
        const uint64_t i = b0 + (uint64_t)threadIdx.x * PGX_SCAN_ITEMS + t;
        vals[t] = i < n ? pgx_scan_load(mode, in, i, min_len) : 0;
        v += vals[t];
    }
    uint64_t tot;
    uint64_t ex = base + pgx_block_excl_scan(v, s_wave, tot);
    for (int t = 0; t < PGX_SCAN_ITEMS; t++) {
        const uint64_t i = b0 + (uint64_t)threadIdx.x * PGX_SCAN_ITEMS + t;
        if (i < n) out[i] = ex;
        ex += vals[t];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[n] = grand;
        if (total_out) *total_out = grand; // a second copy next to other scalars the host reads back together
    }
}

// The same scan in ONE launch (decoupled look-back): tiles of 4096 items take their numbers from a counter in the order they start, publish their
// total, look back over the tiles before them until one has published its inclusive prefix, and publish their own.  A tile word is
// epoch (18 bits) | state (2: 1 = total, 2 = inclusive prefix) | value (44 bits) in one 64-bit store, so a word of an earlier scan over the same
// buffer reads as "nothing yet" and nothing has to be cleared between scans.  Loads and stores are coalesced (a wave scans 64 consecutive items per
// round with lane shifts, sixteen rounds), which the three-launch form above was not: 10 M counts take 177 us there.
// state[0]: tile counter (the last tile sets it back to 0), state[1 + t]: word of tile t.
#define PGX_SCAN1_ROUNDS 16
template <int MODE>
__global__ void __launch_bounds__(256)
pgx_scan_onepass_kernel(const void *in, uint64_t n_cap, uint64_t min_len, uint64_t *__restrict__ out, uint64_t *__restrict__ total_out,
                        const uint64_t *__restrict__ n_dev, unsigned long long *__restrict__ state, uint32_t epoch) {
    constexpr int mode = MODE;
    __shared__ uint64_t s_wave[4];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_tile;
    const uint64_t n = n_dev ? (*n_dev < n_cap ? *n_dev : n_cap) : n_cap;
    if (threadIdx.x == 0) s_tile = atomicAdd(reinterpret_cast<uint32_t *>(state), 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    // tiles behind the one that holds item n - 1 have nothing to do (n may be a device count well below the capacity the grid was sized for): they
    // leave at once, and nobody looks back at them
    const uint32_t last_tile = n ? (uint32_t)((n - 1) / (256u * PGX_SCAN1_ROUNDS)) : 0u;
    if (tile > last_tile) {
        if (tile == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<uint32_t *>(state), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t b0 = (uint64_t)tile * (256u * PGX_SCAN1_ROUNDS) + (uint64_t)w * (64u * PGX_SCAN1_ROUNDS);
    uint64_t x[PGX_SCAN1_ROUNDS], carry = 0;
#pragma unroll
    for (int r = 0; r < PGX_SCAN1_ROUNDS; r++) { // (all sixteen loads first: one memory latency per tile, not one per round)
        const uint64_t i = b0 + (uint64_t)(r * 64 + lane);
        x[r] = i < n ? pgx_scan_load(mode, in, i, min_len) : 0;
    }
#pragma unroll
    for (int r = 0; r < PGX_SCAN1_ROUNDS; r++) {
        const uint64_t v = x[r];
        uint64_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        x[r] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) s_wave[w] = carry;
    __syncthreads();
    uint64_t wbase = 0, tile_total = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { if (i < w) wbase += s_wave[i]; tile_total += s_wave[i]; }
    const unsigned long long ep = (unsigned long long)(epoch & 0x3FFFFu) << 46;
    const unsigned long long vmask = (1ull << 44) - 1ull;
    if (w == 0) {
        uint64_t prefix = 0;
        if (tile != 0) {
            if (lane == 0) __hip_atomic_store(state + 1 + tile, ep | (1ull << 44) | (tile_total & vmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t j = (int64_t)tile - 1;
            for (;;) { // 64 predecessors at a time, nearest first (lane 0 = tile j)
                const int64_t idx = j - lane;
                unsigned long long word = ep | (2ull << 44); // (before tile 0: an inclusive prefix of 0)
                if (idx >= 0) word = __hip_atomic_load(state + 1 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t st = (word >> 46) == (ep >> 46) ? (uint32_t)(word >> 44) & 3u : 0u;
                const unsigned long long ready = __ballot(st != 0u), full = __ballot(st == 2u);
                // the run of published words that starts at lane 0 and ends at the first inclusive prefix (or at lane 63)
                const unsigned long long gap = ~ready;
                const int stop_gap = gap ? (int)__ffsll((long long)gap) - 1 : 64, stop_full = full ? (int)__ffsll((long long)full) - 1 : 64;
                if (stop_full < stop_gap) { // an inclusive prefix before any unpublished tile: sum up to it and stop
                    uint64_t v = lane <= stop_full ? (uint64_t)(word & vmask) : 0;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                    prefix += __shfl(v, 0, 64);
                    break;
                }
                if (stop_gap == 64) { // 64 totals, no prefix among them: take them all and look further back
                    uint64_t v = (uint64_t)(word & vmask);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                    prefix += __shfl(v, 0, 64);
                    j -= 64;
                    continue;
                }
                __builtin_amdgcn_s_sleep(2); // a tile in the window has not published yet
            }
        }
        if (lane == 0) {
            __hip_atomic_store(state + 1 + tile, ep | (2ull << 44) | ((prefix + tile_total) & vmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = prefix;
        }
    }
    __syncthreads();
    const uint64_t P = s_prefix + wbase;
#pragma unroll
    for (int r = 0; r < PGX_SCAN1_ROUNDS; r++) {
        const uint64_t i = b0 + (uint64_t)(r * 64 + lane);
        if (i < n) out[i] = P + x[r];
    }
    if (tile == last_tile && threadIdx.x == 0) {
        const uint64_t grand = s_prefix + tile_total;
        out[n] = grand;
        if (total_out) *total_out = grand;
    }
    if (tile == gridDim.x - 1 && threadIdx.x == 0) // (every tile has its number by now)
        __hip_atomic_store(reinterpret_cast<uint32_t *>(state), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define PGX_SCAN1_INSTANTIATE(M) \
    template __global__ void pgx_scan_onepass_kernel<M>(const void *, uint64_t, uint64_t, uint64_t *, uint64_t *, const uint64_t *, unsigned long long *, uint32_t);
PGX_SCAN1_INSTANTIATE(0) PGX_SCAN1_INSTANTIATE(1) PGX_SCAN1_INSTANTIATE(2) PGX_SCAN1_INSTANTIATE(3) PGX_SCAN1_INSTANTIATE(4) PGX_SCAN1_INSTANTIATE(5)

// ------------------------------------------------------------------------------------------
// MEM compaction: slots (pgx_slot_index: four per read in a dense slot-major array, the rest in the arena / at the read's worst-case offset) -> dense CSR in read order
__global__ void __launch_bounds__(256)
pgx_compact_mems_kernel(uint64_t first_read, uint64_t n_reads, const uint64_t *__restrict__ slot_off, uint64_t slot_base,
                        const pgx_mem *__restrict__ slots, const uint32_t *__restrict__ mem_count,
                        const uint64_t *__restrict__ local_off, uint64_t mem_base, pgx_mem *__restrict__ mems, uint64_t cap_mems,
                        uint64_t *__restrict__ abort, const uint32_t *__restrict__ ovf_base, uint64_t ovf_cap) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // read first_read + t of this chunk
    if (t >= n_reads) return;
    const uint64_t i = first_read + t;
    const uint32_t c = mem_count[i];
    const uint64_t src = c <= PGX_FAST_SLOTS ? 0ull : (ovf_cap ? (uint64_t)ovf_base[i] - PGX_FAST_SLOTS : slot_off[i] - slot_base), dst = mem_base + local_off[t];
    if (dst + c > cap_mems) { // speculative sizing: the MEM array was sized from an earlier run and this one has more
        if (c && abort) atomicOr((unsigned long long *)abort, 16ull);
        return;
    }
    for (uint32_t u = 0; u < c; u++) mems[dst + u] = slots[pgx_slot_index(t, n_reads, src, u)];
}


// ------------------------------------------------------------------------------------------
// query_tags path: FastLocate::count / count_encoded (r-index.hpp:540-556), one lane
// per read: range = {0, n-1}; for each symbol from the end: LF (src/r-index.cpp:650-711).  An empty
// range is {1, 0} and stays empty.
template <bool LDS_IMAGE>
__global__ void __launch_bounds__(256)
pgx_count_kernel(PgxDevImage img, const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets, uint64_t n_reads,
                 pgx_range *__restrict__ out) {
    __shared__ uint32_t s_cnt[256];
    __shared__ uint64_t s_C[8];
    PGX_LDS_CARVE(img);
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_cnt[i] = img.consts->cnt_tab[i];
    if (threadIdx.x < 8) s_C[threadIdx.x] = img.consts->C[threadIdx.x];
    if (LDS_IMAGE) {
        const uint32_t nb4 = img.n_blocks * 4;
        for (uint32_t i = threadIdx.x; i < nb4; i += blockDim.x) lds_blocks[img.dense ? (i >> 2) * PGX_DENSE_LDS_U4 + (i & 3u) : i] = img.blocks[i];
        if (!img.dense)
            for (uint64_t i = threadIdx.x; i < img.dir_entries; i += blockDim.x) lds_dir[i] = img.dir[i];
        if (!img.dense)
            for (uint32_t i = threadIdx.x; i < img.n_blocks; i += blockDim.x) lds_blow[i] = img.blow[i];
    }
    __syncthreads();
    const uint64_t rid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rid >= n_reads) return;
    const uint64_t base = offsets[rid], len = offsets[rid + 1] - base;
    uint64_t lo = 0, hi = img.n - 1;
    if (img.n == 0) { lo = 1; hi = 0; }
    for (uint64_t i = len; i > 0 && lo <= hi; i--) {
        const uint32_t e = s_cnt[reads[base + i - 1]];
        if (PGX_EXT_KILL(e)) { lo = 1; hi = 0; break; }
        uint64_t A0, A1, dB;
        pgx_rank_pair<LDS_IMAGE>(img, lds_blocks, lds_dir, lds_blow, lo, hi + 1, PGX_EXT_CV(e), 0u, A0, A1, dB);
        if (A1 == A0) { lo = 1; hi = 0; break; } // sym_inside == 0 -> {1, 0}
        lo = A0 + s_C[PGX_EXT_V(e)];
        hi = lo + (A1 - A0) - 1;
    }
    pgx_range r;
    r.first = lo; r.second = hi;
    out[rid] = r;
}
template __global__ void pgx_count_kernel<false>(PgxDevImage, const uint8_t *, const uint64_t *, uint64_t, pgx_range *);
template __global__ void pgx_count_kernel<true>(PgxDevImage, const uint8_t *, const uint64_t *, uint64_t, pgx_range *);

// One LF step per query (FastLocate::LF src/r-index.cpp:650-687 / LF_encoded :689-711): the inclusive range [first, second]
// mapped by `sym`; an empty input or result is {1, 0}.  Same tables as pgx_count_kernel.
__global__ void __launch_bounds__(256)
pgx_lf_kernel(PgxDevImage img, const pgx_range *__restrict__ in, const uint8_t *__restrict__ sym, uint64_t n, pgx_range *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = img.consts->cnt_tab[sym[i]];
    pgx_range r;
    r.first = 1; r.second = 0;
    const uint64_t lo = in[i].first, hi = in[i].second;
    if (!PGX_EXT_KILL(e) && lo <= hi) { // positions beyond n rank like n (predecessor -> last block, totals)
        uint64_t A0, A1, dB;
        pgx_rank_pair<false>(img, nullptr, nullptr, nullptr, lo, hi + 1, PGX_EXT_CV(e), 0u, A0, A1, dB);
        if (A1 != A0) {
            r.first = A0 + img.consts->C[PGX_EXT_V(e)];
            r.second = r.first + (A1 - A0) - 1;
        }
    }
    out[i] = r;
}

// find_mems_function for n independent (read, start) pairs, one lane each (the compat header's per-call entry point and the
// tests of the state machine; the batch kernel above is the product path).
__global__ void __launch_bounds__(256)
pgx_fmf_kernel(PgxDevImage img, const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ read_of,
               const uint64_t *__restrict__ xs, uint64_t n, uint64_t min_len, uint64_t min_occ, PgxHeavyResult *__restrict__ out) {
    __shared__ uint32_t s_ext[512];
    __shared__ uint64_t s_C[8];
    pgx_stage_tables<false>(img, s_ext, s_C, nullptr, nullptr, nullptr);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t rid = read_of[i], base = offsets[rid];
    const int32_t len = (int32_t)(offsets[rid + 1] - base);
    PgxHeavyResult r;
    if (xs[i] > (uint64_t)len) { // undefined in the reference (len - x wraps, :658); defined here as "return len"
        r.mem.start = xs[i]; r.mem.end = 0; r.mem.bwt_start = 0; r.mem.size = 0;
        r.next_x = (uint32_t)len; r.n_ext = 0; r.has_mem = 0; r.pad = 0;
    } else {
        r = pgx_fmf_eval<false>(img, nullptr, nullptr, nullptr, s_ext, s_C, reads + base, len, (int32_t)xs[i], min_len, min_occ);
    }
    out[i] = r;
}

// ------------------------------------------------------------------------------------------
// FastLocate::rankAt_encoded as the reference executes it on an encoded index without N (quirk 3, pgx_device.h PgxLitImage)
__device__ __forceinline__ uint64_t pgx_lit_rank(const PgxLitImage &lit, uint64_t pos, uint32_t target) {
    // predecessor: last block start <= pos (positions at or beyond the end fall into the last block)
    uint64_t lo = 0, hi = lit.n_blocks; // bstart[0] = 0
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (lit.bstart[mid] <= pos) lo = mid; else hi = mid;
    }
    const uint64_t rel = pos - lit.bstart[lo];
    uint64_t rank = 0, cur = 0;
    for (uint32_t e = lit.roff[lo]; e < lit.roff[lo + 1]; e++) { // EncodedBlock::rank_of_code, src/r-index.cpp:114-131
        const uint64_t u = lit.runs[e], len = u & ((1ull << 56) - 1);
        if ((uint32_t)(u >> 56) == target) {
            if (cur + len > rel) { rank += rel - cur; break; }
            rank += len;
        }
        cur += len;
        if (cur > rel) break;
    }
    return rank + lit.cum[lo * 6 + target];
}
__global__ void __launch_bounds__(256)
pgx_lit_count_kernel(PgxLitImage lit, const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets, const pgx_range *__restrict__ in,
                     const uint8_t *__restrict__ sym, uint64_t n, pgx_range *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t lo, hi, len, base = 0;
    if (in) { lo = in[i].first; hi = in[i].second; len = 1; }
    else { lo = 0; hi = lit.n - 1; base = offsets[i]; len = offsets[i + 1] - base; }
    for (uint64_t t = len; t > 0; t--) { // LF_encoded, src/r-index.cpp:689-711: no symbol is rejected, an empty range stays {1, 0}
        const uint32_t byte = in ? sym[i] : reads[base + t - 1];
        if (lo > hi) { lo = 1; hi = 0; continue; }
        const uint32_t target = lit.code_of[byte];
        const uint64_t f = pgx_lit_rank(lit, lo, target), inside = pgx_lit_rank(lit, hi + 1, target) - f;
        if (inside == 0) { lo = 1; hi = 0; continue; }
        lo = f + lit.C[lit.cslot_of[byte]];
        hi = lo + inside - 1;
    }
    pgx_range r;
    r.first = lo; r.second = hi;
    out[i] = r;
}

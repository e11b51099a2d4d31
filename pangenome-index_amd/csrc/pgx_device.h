// pgx_device.h -- kernel-side view of the device image + kernel declarations (HIP only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pgx.h"
#include "pgx_image.h"

// Counters of one pgx_batch_run: u64 slots of pgx_batch::counters (the find_mems kernels receive the base pointer as `n_ext_total`).
// Slots below 16 are cleared per chunk of reads and per attempt; the others once per run.
enum PgxCounterSlot {
    PGX_CTR_EXT = 0,          // extensions performed (equal to the oracle's count)
    PGX_CTR_TAG_OVERFLOW = 1, // tag queries that read past the stored runs (quirk 7)
    PGX_CTR_CURSOR = 5,       // read cursor of the main launch
    PGX_CTR_HEAVY = 8,        // reads handed to pgx_find_mems_heavy_kernel
    PGX_CTR_OVF32 = 9,        // a coordinate left 32 bits (NARROW kernels): the chunk is repeated in 64 bits
    PGX_CTR_MEMS = 10,        // MEMs of the chunk (scan total)
    PGX_CTR_REDO = 11,        // extensions the pairs kernel took through the image it accompanies (flagged blocks, wide intervals)
    PGX_CTR_REDO_CURSOR = 12, // (unused since the pairs kernel no longer hands reads on)
    PGX_CTR_SIDE_CURSOR = 13, // read cursor of the side-stream launch (reads with a byte outside A C G T)
    PGX_CTR_OVF_TOP = 14,     // slots handed out of the arena of fifth-and-later MEMs (pgx_slot_extent)
    PGX_CTR_OVF_ABORT = 15,   // the arena was too small: the chunk is repeated in the worst-case slot layout
    PGX_CTR_TAG0 = 16,        // 16..24: tag stage (tag_pipeline)
    PGX_CTR_ABORT = 30,       // abort flag of a speculative run
    // what the kernels ask of the memory system (always on; bench.py's roofline): lane trips that fetch a rank-image line that is
    // not one every lane shares (a stage's first trip probes the full interval: block 0 / the last block), seed / end table entries read
    PGX_CTR_PAIRS_LINES = 32, // pgx_find_mems_pairs_kernel: 128-byte block lines (second-block trips included)
    PGX_CTR_PAIRS_SEEDS = 33, //                             seed / end table entries (16 bytes each, one line each)
    PGX_CTR_FM_LINES = 34,    // pgx_find_mems_kernel (global-memory images): 128-byte lines holding the blocks of its probes
    PGX_CTR_FM_SEEDS = 35,
    PGX_CTR_PAIRS_TWO = 36,   // pairs kernel: lane trips that performed two extensions
    // -DPGX_FM_STATS builds only (scripts/fm_stats.sh)
    PGX_CTR_ST_TRIPS = 40, PGX_CTR_ST_LIVE = 41, PGX_CTR_ST_LONGEST = 42,                  // pgx_find_mems_kernel: wave trips, live lane trips, longest wave
    PGX_CTR_ST_PAIR_TRIPS = 43, PGX_CTR_ST_PAIR_LIVE = 44, PGX_CTR_ST_PAIR_WAIT = 45, PGX_CTR_ST_PAIR_FRESH = 46,
    PGX_CTR_ST_PAIR_T_REFILL = 47, PGX_CTR_ST_PAIR_T_TOTAL = 48, PGX_CTR_ST_PAIR_REFILLS = 49, PGX_CTR_ST_PAIR_T_SEED = 50, PGX_CTR_ST_PAIR_T_LINE = 51, // clock ticks (s_memtime) of the waves in the refill loop / in all, refill rounds, ticks waiting for the seed entry / for the block line
    PGX_CTR_SLOTS = 64,  // (what the host reads back)
    PGX_CTR_ARENA0 = 64, // counters of the PGX_ARENA_SUBS sub-arenas, 16 slots (one cache line) apart
    PGX_CTR_ALL = 64 + 16 * 64
};
#define PGX_ARENA_SUBS 64u
__global__ void pgx_arena_demand_kernel(unsigned long long *ctr);
#define PGX_TBUCKET_RUNS 10u // run starts a bucket line holds (pgx_tag_bucket_kernel); a bucket with more is flagged and answered through tdir / tpair

#define PGX_DENSE_LDS_U4 5 // uint4 slots per dense block in LDS (64 data bytes + 16 of padding)
#define PGX_FM_THREADS 256
#define PGX_LCE_MAX_OCC 128u // widest interval whose forward stage goes through the text (occurrence indexes and counts are bytes of the lane's state word)
#define PGX_FM_WAVES_PER_SIMD 4 // __launch_bounds__ 2nd argument: caps the kernel at 128 VGPRs

// passed by value to every kernel (all pointers are device pointers)
struct PgxDevImage {
    const uint4 *blocks;      // n_blocks * 4 (64-byte rank blocks)
    const uint64_t *dir;      // dir_entries (64-bit entries, see pgx_image.h)
    const uint16_t *blow;     // n_blocks (low dir_shift bits of each block start)
    const PgxConsts *consts;  // tables (ext_tab, C, slot_code)
    const uint64_t *tstart;   // n_tag_runs
    const uint64_t *tvals;    // n_tag_items
    const uint32_t *tdir;     // tag_dir_entries
    const ulonglong2 *tpair;  // (tstart[r], tvals[r]) side by side, max(n_tag_runs, n_tag_items) entries (NULL: not built)
    const uint4 *tbucket;     // tag runs by bucket of 2^tbucket_shift BWT positions, one 128-byte line each (pgx_tag_kernels.hip; NULL: not built)
    uint64_t n_tbuckets;
    uint32_t tbucket_shift;
    uint64_t n;
    uint64_t dir_entries;
    uint64_t n_tag_runs, n_tag_items, tag_dir_entries;
    uint32_t n_blocks;
    uint32_t dir_shift;
    uint32_t excl_mask;
    uint32_t tag_dir_shift;
    uint32_t dense; // image kind: 0 run-length blocks, 1 dense bit-plane blocks (PGX_IMAGE_DENSE), 2 dense2 (PGX_IMAGE_DENSE2); dir / blow unused unless 0
    const uint32_t *exc; // dense2: exception runs
    // k-mer seed table (dense images in global memory): entry idx = bi-interval after backward-extending the full interval by
    // the k bytes of a window, last byte first (pgx_seed_build_kernel); 0 = no table
    uint32_t seed_k;
    const uint4 *seed; // 4^seed_k entries {k lo, k' lo, s lo, k hi | k' hi << 8 | s hi << 16 | depth << 24}
    // end table: the same for a stage that starts at j = len: the full interval extended by 0 (what pattern[len] reads as) and then by the
    // seed_end_k bytes before the end of the read; depths count the extension by 0
    uint32_t seed_end_k;
    const uint4 *seed_end;
    // host side only (pgx_batch_run puts one of the two tables into seed / seed_k of the copy it launches with): the first table and a shallower one
    uint32_t seed_k_main, seed_k_small;
    const uint4 *seed_main, *seed_small;
    // PAIRS image (NULL without one): blocks, first extensions of the full interval
    const uint4 *pairs;
    const uint4 *first_ext;
    uint32_t pair_runs;
    // WIDE images (pgx_image.h): superblock bases and shifts; dense = 3 then (dense2 blocks with delta counts, 64-bit positions)
    const uint64_t *sbase2, *pbase;
    uint32_t d2_sb_shift, pairs_sb_shift, n_sb2, n_sbp;
    uint32_t wide;
    uint32_t pairs_stride; // positions between PAIRS block starts: PGX_PAIRS_SYMS or PGX_PAIRS_STRIDE64 (pgx_image.h)
    // suffix array + text (NULL without; narrow images with a PAIRS image only: pgx_image.h "LCE image"): the forward stage of a MEM whose interval is
    // narrow is finished by comparing the read with the text at the interval's occurrences, one occurrence per trip, instead of two symbols per trip
    const uint32_t *lce_sa;    // n entries: position of suffix i in lce_text
    const uint32_t *lce_text;  // two bits per symbol (A C T G = 0 1 2 3, the order of the packed reads), 16 symbols per word, sequences with their endmarkers
    const uint32_t *lce_flags; // one bit per 128-byte line of lce_text: the line holds a symbol outside A C G T or lies behind the text
    const uint8_t *lce_lcp;    // n entries (NULL without): symbols suffix i shares with suffix i - 1, capped at PGX_LCP_CAP; PGX_LCP_UNKNOWN where the comparison met a flagged line
    uint32_t lce_max;          // widest interval that goes this way
    uint32_t refill_min;       // LCE kernel: idle lanes of a wave wait for this many before the wave fetches new reads
};
#define PGX_LCP_CAP 254u     // lce_lcp: "this many symbols or more"
#define PGX_LCP_UNKNOWN 255u // lce_lcp: not known (the comparison touched a flagged line of the text)
#define PGX_SEED_UNUSABLE 255u // depth value of entries the kernels must not use (a coordinate does not fit the entry)
#define PGX_SEED_MAX_K 16
#define PGX_SEED_SMALL_K 10 // depth of the second table (searches whose min_len is below the depth of the first)

// heavy reads (pgx_kernels.hip): handed from pgx_find_mems_kernel to pgx_find_mems_heavy_kernel
struct pgx_heavy_item {
    uint64_t rid;
    uint32_t x, nm; // next start position, MEMs already written
};
struct PgxHeavyResult { // find_mems_function(x) of one start position
    pgx_mem mem;
    uint32_t next_x, n_ext, has_mem, pad;
};
#define PGX_FM_HEAVY_EXT 2048u    // extensions spent on one read before the rest is handed on (ordinary 150-bp reads: ~200)
#define PGX_FM_SIDE_HEAVY_EXT 2048u // the same for the launch on the second stream (reads with a byte outside A C G T); see pgx_batch_run
#define PGX_FM_HEAVY_MAXLEN 4096u // reads up to this length take part (per-start results live in scratch)
#define PGX_FM_HEAVY_CAP 8192u    // heavy reads per launch; beyond that lanes simply continue sequentially
#define PGX_FM_HEAVY_GRID 64u
template <bool LDS_IMAGE>
__global__ void pgx_find_mems_heavy_kernel(PgxDevImage img, const uint8_t *reads, const uint64_t *offsets, uint64_t min_len, uint64_t min_occ,
                                           const uint64_t *slot_off, uint64_t slot_base, pgx_mem *slots, uint32_t *mem_count,
                                           unsigned long long *n_ext_total, const pgx_heavy_item *heavy_list,
                                           const unsigned long long *heavy_count, uint32_t heavy_cap, PgxHeavyResult *scratch, uint64_t chunk_first,
                                           uint64_t chunk_reads, uint32_t *ovf_base, uint64_t ovf_cap);

__global__ void pgx_seed_build_kernel(PgxDevImage img, const uint4 *src, uint4 *dst, uint32_t level, uint64_t n_dst, uint64_t limit, int end_table);
template <bool LDS_IMAGE, int DENSE, bool NARROW, bool SEED> // DENSE = image kind
__global__ void pgx_find_mems_kernel(PgxDevImage img, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads,
                                     uint64_t min_len, uint64_t min_occ, const uint64_t *slot_off, pgx_mem *slots,
                                     uint32_t *mem_count, unsigned long long *n_ext_total, unsigned long long *cursor, uint64_t first_read,
                                     uint64_t slot_base, uint32_t heavy_ext, uint32_t heavy_cap, pgx_heavy_item *heavy_list, unsigned long long *heavy_count,
                                     const pgx_heavy_item *rid_list, const unsigned long long *rid_count, uint32_t *ovf_base, uint64_t ovf_cap);
// PAIRS image (pgx_image.h): two extensions per loop trip; an extension its blocks cannot answer (special positions) goes through the other image
template <bool SEED, bool WIDE, bool PACKED, bool COOP, bool S64, bool LCE>
__global__ void pgx_find_mems_pairs_kernel(PgxDevImage img, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads,
                                           uint64_t min_len, uint64_t min_occ, const uint64_t *slot_off, pgx_mem *slots,
                                           uint32_t *mem_count, unsigned long long *n_ext_total, unsigned long long *cursor, uint64_t first_read,
                                           uint64_t slot_base, uint32_t heavy_ext, uint32_t heavy_cap, pgx_heavy_item *heavy_list, unsigned long long *heavy_count,
                                           const uint8_t *skip, const uint32_t *packed, uint32_t pk_words,
                                           uint32_t *ovf_base, uint64_t ovf_cap);
__global__ void pgx_bad_chunks_kernel(const uint8_t *reads, uint64_t n_bytes, uint64_t *chunks, unsigned long long *count, uint64_t cap, uint32_t *packed);
__global__ void pgx_unpack_reads_kernel(const uint32_t *packed, uint64_t n_chunks, uint8_t *reads);
__global__ void pgx_side_reads_kernel(uint8_t *reads, const uint64_t *offsets, const uint64_t *side_ids, const uint64_t *side_off, const uint8_t *side_bytes, uint64_t n_side,
                                      uint8_t *flags, pgx_heavy_item *list, unsigned long long *count);
__global__ void pgx_classify_reads_kernel(const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads, const uint64_t *chunks, const unsigned long long *n_chunks,
                                          uint64_t cap, uint32_t *flag_words, pgx_heavy_item *list, unsigned long long *count);
__global__ void pgx_first_ext_kernel(PgxDevImage img, uint4 *out); // out[byte] = {k, k', s, 0} of the full interval extended backward by byte; out[256 + byte]: by 0, then by byte
__global__ void pgx_rank_kernel(PgxDevImage img, const uint64_t *pos, uint64_t n, int true_codes, uint64_t *out);
template <bool LOOP, bool MULHI> __global__ void pgx_rank_probe_kernel(PgxDevImage img, const uint64_t *pos, uint64_t n, uint64_t *out);
template <bool LDS_IMAGE>
__global__ void pgx_extend_kernel(PgxDevImage img, const pgx_biint *in, const uint8_t *sym, const uint8_t *forward, uint64_t n,
                                  pgx_biint *out);
template <bool LDS_IMAGE>
__global__ void pgx_count_kernel(PgxDevImage img, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads, pgx_range *out);
__global__ void pgx_fmf_kernel(PgxDevImage img, const uint8_t *reads, const uint64_t *offsets, const uint64_t *read_of, const uint64_t *xs, uint64_t n,
                               uint64_t min_len, uint64_t min_occ, PgxHeavyResult *out);
__global__ void pgx_lf_kernel(PgxDevImage img, const pgx_range *in, const uint8_t *sym, uint64_t n, pgx_range *out);
// n_dev (may be NULL): the element count lives on the device, `n` is then the capacity the launch was sized for
__global__ void pgx_scan_partial_kernel(int mode, const void *in, uint64_t n, uint64_t min_len, uint64_t *block_sums, const uint64_t *n_dev);
template <int MODE>
__global__ void pgx_scan_onepass_kernel(const void *in, uint64_t n, uint64_t min_len, uint64_t *out, uint64_t *total_out, const uint64_t *n_dev,
                                        unsigned long long *state, uint32_t epoch);
__global__ void pgx_scan_sums_kernel(uint64_t *block_sums, uint64_t nb);
__global__ void pgx_scan_apply_kernel(int mode, const void *in, uint64_t n, uint64_t min_len, const uint64_t *block_sums,
                                      uint64_t nb, uint64_t *out, uint64_t *total_out, int raw_sums, const uint64_t *n_dev);
__global__ void pgx_compact_mems_kernel(uint64_t first_read, uint64_t n_reads, const uint64_t *slot_off, uint64_t slot_base,
                                        const pgx_mem *slots, const uint32_t *mem_count, const uint64_t *local_off,
                                        uint64_t mem_base, pgx_mem *mems, uint64_t cap_mems, uint64_t *abort, const uint32_t *ovf_base, uint64_t ovf_cap);
__global__ void pgx_tag_pair_kernel(const uint64_t *tstart, const uint64_t *tvals, uint64_t n_runs, uint64_t n_items, ulonglong2 *out);
__global__ void pgx_tag_bucket_kernel(const uint64_t *tstart, const uint64_t *tvals, uint64_t n_runs, uint64_t n_items, uint32_t shift, uint64_t n_buckets, uint4 *out);
#define PGX_TAG_LOCATE_THREADS 1024 // workgroup of pgx_tag_locate_kernel (one list atomic per workgroup)
#define PGX_SORT_LDS_CAP 2048 // values per wave sorted in an LDS slice (pgx_tag_sort_unique_kernel)
#define PGX_TAG_SMALL 16      // queries with at most this many runs take the 16-lane path
#define PGX_SORT_WG_LDS_CAP 16384 // values one workgroup sorts in (dynamic) LDS (pgx_tag_sort_large_kernel)

// every kernel of the tag stage: (count as a value = capacity, device pointer to the actual count or NULL, abort flag or NULL)
__global__ void pgx_tag_locate_kernel(PgxDevImage img, const pgx_mem *mems, const uint64_t *qstart, const uint64_t *qend,
                                      uint64_t n, const uint64_t *n_dev, const uint64_t *abort, uint64_t *run_nums, uint64_t *first_item, uint64_t *need,
                                      uint64_t *big_list, uint64_t *large_list, unsigned long long *n_big, unsigned long long *n_large, uint64_t *single,
                                      uint64_t *ucount, unsigned long long *n_overflow, uint64_t *small_list, unsigned long long *n_small);
__global__ void pgx_tag_small_kernel(PgxDevImage img, const uint64_t *list, uint64_t n, const uint64_t *n_dev, const uint64_t *abort, const uint64_t *run_nums,
                                     const uint64_t *first_item, const uint64_t *seg_off, uint64_t *buf, uint64_t *ucount, unsigned long long *n_overflow);
__global__ void pgx_tag_gather_kernel(PgxDevImage img, const uint64_t *list, uint64_t n_list, const uint64_t *n_dev, const uint64_t *abort,
                                      const uint64_t *run_nums, const uint64_t *first_item, const uint64_t *seg_off, uint64_t *buf,
                                      unsigned long long *n_overflow);
__global__ void pgx_tag_sort_unique_kernel(const uint64_t *list, uint64_t n_list, const uint64_t *n_dev, const uint64_t *abort, const uint64_t *run_nums,
                                           const uint64_t *seg_off, uint64_t *buf, uint64_t *ucount);
__global__ void pgx_tag_sort_large_kernel(const uint64_t *list, uint64_t n_list, const uint64_t *n_dev, const uint64_t *abort, const uint64_t *run_nums,
                                          const uint64_t *seg_off, uint64_t *buf, uint64_t *scratch, const uint64_t *scratch_off, uint64_t *ucount);
__global__ void pgx_tag_dedup_kernel(const uint64_t *list, uint64_t n_list, const uint64_t *n_dev, const uint64_t *abort, const uint64_t *first_item,
                                     const uint64_t *run_nums, unsigned long long *table, uint64_t table_mask, uint64_t *reps, unsigned long long *n_rep,
                                     uint64_t *pairs, unsigned long long *n_dup);
__global__ void pgx_tag_copy_dups_kernel(const uint64_t *pairs, uint64_t n_pairs, const uint64_t *n_dev, const uint64_t *abort, uint64_t n_tag_items,
                                         const uint64_t *first_item, const uint64_t *run_nums, uint64_t *seg_off, uint64_t *buf, uint64_t *ucount,
                                         unsigned long long *n_overflow);
__global__ void pgx_tag_compact_kernel(const uint64_t *list, uint64_t n, const uint64_t *n_dev, const uint64_t *abort, const uint64_t *ucount,
                                       const uint64_t *seg_off, const uint64_t *buf, const uint64_t *pos_off, uint64_t *positions, uint64_t max_count);
__global__ void pgx_tag_compact_single_kernel(uint64_t n, const uint64_t *n_dev, const uint64_t *abort, const uint64_t *run_nums, const uint64_t *single,
                                              const uint64_t *pos_off, uint64_t *positions);
__global__ void pgx_tag_compact_list_kernel(const uint64_t *list, uint64_t n_list, const uint64_t *n_dev, const uint64_t *abort, const uint64_t *ucount,
                                            const uint64_t *seg_off, const uint64_t *buf, const uint64_t *pos_off, uint64_t *positions, uint64_t max_count);
__global__ void pgx_spec_check_kernel(const uint64_t *v0, uint64_t c0, const uint64_t *v1, uint64_t c1, const uint64_t *v2, uint64_t c2,
                                      const uint64_t *v3, uint64_t c3, uint64_t *abort);
#define PGX_TAG_COMPACT_SMALL 256 // segments up to this many unique values are copied by the 16-lane kernel

// locate image (pgx_image.h), passed by value to the locate kernels
struct PgxLocImage {
    const uint64_t *rstart, *rsamp; // n_runs + 1, n_runs
    const uint32_t *rdir;
    const uint64_t *lpos, *lnext;   // n_last each
    const uint32_t *ldir;
    uint64_t n, n_runs, n_last, max_length, rdir_entries, ldir_entries;
    uint32_t rdir_shift, ldir_shift;
};
__global__ void pgx_locate_next_kernel(PgxLocImage loc, const uint64_t *prev, uint64_t n, uint64_t *out);
// LCE image (pgx_image.h): sequence lengths from the suffixes that start with an endmarker, suffix array in text coordinates + the text's symbols
// (the first symbol of suffix i is the one whose C-bucket holds i), text packed to two bits + line flags
__global__ void pgx_lce_seqlen_kernel(const uint64_t *sa, uint64_t n_seq, uint64_t max_length, unsigned long long *seq_len);
__global__ void pgx_lce_scatter_kernel(const uint64_t *sa, uint64_t n, uint64_t max_length, const uint64_t *seq_start, uint64_t n_seq, uint64_t c1, uint64_t c2, uint64_t c3,
                                       uint64_t c4, uint64_t c5, uint32_t *sa32, uint8_t *text8, unsigned long long *bad);
__global__ void pgx_lce_pack_kernel(const uint8_t *text8, uint64_t n, uint64_t n_words, uint32_t *text32, uint32_t *flags);
__global__ void pgx_lce_lcp_kernel(const uint32_t *sa32, const uint32_t *text32, const uint32_t *flags, uint64_t n, uint8_t *lcp);
__global__ void pgx_lce_rc_check_kernel(const uint8_t *text8, const uint64_t *seq_start, uint64_t n_seq, uint64_t n, unsigned long long *bad);
__global__ void pgx_locate_plan_kernel(PgxLocImage loc, const uint64_t *qs, const uint64_t *qe, uint64_t n, uint64_t *run0,
                                       uint64_t *n_pieces);
__global__ void pgx_locate_walk_kernel(PgxLocImage loc, const uint64_t *qs, const uint64_t *qe, uint64_t n_queries,
                                       const uint64_t *run0, const uint64_t *piece_off, uint64_t n_pieces, const uint64_t *val_off,
                                       int seq_ids, uint64_t *out);

// literal count image: COMPAT count_encoded / LF_encoded on an encoded index without N, block by block as the
// reference's rankAt_encoded (src/r-index.cpp:570-590) sees it -- true cumulative counts, run scan one varint late
struct PgxLitImage {
    const uint64_t *bstart, *cum, *runs; // n_blocks ; 6 n_blocks ; all runs (code << 56 | length)
    const uint32_t *roff;                // n_blocks + 1
    const uint32_t *code_of, *cslot_of;  // 256 each
    uint64_t C[8];
    uint64_t n, n_blocks;
};
// out[i] = LF over `steps` symbols: count mode (in == NULL): the whole read i from {0, n - 1}; LF mode: one symbol sym[i] from in[i]
__global__ void pgx_lit_count_kernel(PgxLitImage lit, const uint8_t *reads, const uint64_t *offsets, const pgx_range *in, const uint8_t *sym,
                                     uint64_t n, pgx_range *out);

// merge_tags passes (pgx_merge_kernels.hip)
__global__ void pgx_mt_file_of_kernel(const uint64_t *da, uint64_t n, uint64_t n_seq, const uint32_t *seq_to_file, uint32_t n_files,
                                      uint8_t *file_of, unsigned long long *n_bad);
__global__ void pgx_mt_expand_kernel(const uint64_t *start, const uint64_t *val, uint64_t n_runs, uint64_t *out);
__global__ void pgx_mt_gather_kernel(const uint8_t *file_of, uint32_t f, const uint64_t *rank, const uint64_t *expanded, uint64_t total,
                                     uint64_t n, uint64_t *tags);
__global__ void pgx_mt_flags_kernel(const uint64_t *tags, uint64_t n, uint64_t n_seq, uint8_t *flags);
__global__ void pgx_mt_compact_kernel(const uint64_t *tags, const uint8_t *flags, const uint64_t *idx, uint64_t n, uint64_t n_seq,
                                      uint64_t *out_val, uint64_t *out_start);

#define PGX_SCAN_BLOCK_ITEMS 2048 // 256 threads x 8 items (pgx_kernels.hip PGX_SCAN_ITEMS)
#define PGX_SCAN1_TILE_ITEMS 4096 // 256 threads x 16 rounds (pgx_scan_onepass_kernel)

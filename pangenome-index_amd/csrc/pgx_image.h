/*
 * pgx_image.h -- flat, pointer-free device image of the FastLocate rank structure and the
 * TagArray, shared by the host builder (pgx_index.cpp) and the HIP kernels (pgx_kernels.hip).
 *
 * Only rank *values* are observable through the reference API, so the layout is
 * chosen for the GPU, not translated from the reference's Elias-Fano + varint blocks:
 *
 *  rank block (64 bytes = one half cache line, 16 dwords, loaded as 4 x dwordx4 by one lane)
 *    dw 0..5   low 32 bits of the six absolute counts c6[code] of BWT[0, block_start)
 *              (code order = nuc = \n A C G N T, include/pangenome_index/utils.hpp:11)
 *    dw 6      high 8 bits of c6[0..3]  (counts are 40-bit: n < 2^40)
 *    dw 7      bits 0..15 high 8 bits of c6[4..5]; bits 16..20 number of run entries used
 *    dw 8..15  16 run entries of 16 bits: (3 * code) << 12 | length (1..4095); length 0 = unused.  The
 *              stored 4-bit field is the bit offset of the code's 3-bit weight in the per-extension
 *              weight rows, so a decode step is two bit-field extracts and two multiply-adds per sum
 *    block_start = sum of the counts whose code is not in `excl_mask` (see PgxConsts)
 *  directory   u64 dir[i], one bucket of 2^dir_shift positions each, about one bucket per two blocks:
 *                bits  0..31  lo  = number of blocks whose start is < (i << dir_shift)
 *                bits 32..39  cnt = blocks starting inside bucket i (saturates at 255)
 *                bits 40..51  low part of the 1st, bits 52..63 of the 2nd block start in the bucket
 *              the block holding pos is lo - 1 + #{blocks of the bucket with low <= pos & mask}: one
 *              8-byte load answers it when cnt <= 2 (the common case); (n >> dir_shift) + 2 entries
 *  block lows  u16 blow[b] = block_start & ((1 << dir_shift) - 1), searched only when cnt > 2
 *              (dir_shift <= 12 so that a low part fits the 12-bit fields)
 *
 *  DENSE rank image (image_kind = PGX_IMAGE_DENSE): for indexes whose BWT is short enough to be stored uncompressed
 *  (64 bytes per 64 symbols = n bytes), the BWA-style layout: block b covers BWT[64 b, 64 b + 64), no directory:
 *    dw 0..7   the same six 40-bit counts as above (counts of BWT[0, 64 b))
 *    dw 8..13  three 64-bit planes: bit i of plane p = bit p of the nuc code of BWT[64 b + i]
 *  rank = header count + popcounts of plane combinations under a prefix mask; both probes of an extension are two
 *  independent 64-byte loads (no dependent directory load) and ~40 ALU operations each instead of a 16-entry run scan.
 *  Not available when a header slot carries the legacy quirk value (excl_mask != 0: blocks must then refine the
 *  reference's 10-run blocks).  (n >> 6) + 1 blocks, so position n has a block of its own or shares the last one.
 *
 *  tag image: u64 tstart[r] (first BWT position of tag run r, ascending), u64 tvals[r] (the
 *  graph position the reference prints: node << 11 | rev << 10 | offset), u32 tdir like dir.
 *
 * The reference's quirks are carried by the tables in PgxConsts, never by branches:
 * the same kernels serve PGX_MODE_COMPAT and PGX_MODE_STRICT.
 */
#ifndef PGX_IMAGE_H
#define PGX_IMAGE_H

#include <stdint.h>

#define PGX_BLOCK_BYTES 64
#define PGX_BLOCK_RUNS 16
#define PGX_RUN_LEN_BITS 12
#define PGX_RUN_LEN_MAX 4095u
#define PGX_COUNT_BITS 40
#define PGX_DIR_MAX_SHIFT 12
#define PGX_IMAGE_RL 0u
#define PGX_IMAGE_DENSE 1u
#define PGX_IMAGE_DENSE2 2u
/* DENSE2 rank image (image_kind = PGX_IMAGE_DENSE2, BWTs shorter than 2^32): 384 symbols per 128-byte block = one cache
 * line serves a rank probe, n / 3 bytes in all (a chr22-scale BWT of 640 M symbols: 213 MB, which stays resident in the
 * 256 MB memory-side cache; the 64-byte-per-64-symbols layout above needs 640 MB and is served from HBM).  A probe reads
 * the 32-byte header and ONE 32-byte sub-block (64 bytes, like a dense block):
 *   dw 0..4    counts of A C G T N in BWT[0, 384 b)  (32 bits each; the count of \n is 384 b minus their sum)
 *   dw 5       exception runs of the block: first one in exc[] (bits 0..23), how many (bits 24..31)
 *   dw 6, 7    64 bits: in-block counts before symbol 128 (bits 0..26) and before symbol 256 (bits 27..53), each three
 *              9-bit fields: symbols with plane-0 bit set, with plane-1 bit set, with both
 *   sub-block s (s = 0, 1, 2; bytes 32 + 32 s ..): 4 dwords of plane 0, 4 dwords of plane 1 over symbols [128 s, 128 s + 128):
 *              bit i of a plane = bit of the 2-bit symbol code, A = 0, C = 1, G = 2, T = 3; the rare other symbols
 *              (\n, N) are stored as 0 and listed in exc[]
 *   exc[e]     start within the block (bits 0..8) | run length (bits 9..17) | kind (bit 18: 0 = \n, 1 = N)
 * rank = header count + sub-block count + popcounts of plane combinations under a prefix mask, corrected by the
 * exception runs of the block. */
#define PGX_D2_SYMS 384u
#define PGX_D2_BLOCK_BYTES 128u
/* PAIRS image (next to a DENSE or DENSE2 image; BWTs shorter than 2^32 whose extension tables are the textbook ones): a two-step FM
 * index.  Position p carries the PAIR (c1, c2) = (BWT[p], BWT[LF(p)]) = the two text symbols before suffix p, so the ranks of
 * pairs at the two ends of an interval give the interval after TWO extensions -- the positions with c1 = a are mapped by LF,
 * in order, onto the interval after the first extension, and BWT there is c2 -- from the same cache line that answers one:
 * find_mems performs half the dependent line fetches.  96 symbols per 128-byte block, 4 n / 3 bytes in all:
 *   dw 4 y + x   (y, x in A C G T = 0..3) number of positions q < 96 b with c1(q) = y, c2(q) = x
 *   dw 16 + y    bits 0..23: number of positions q < 96 b with c1(q) = y and c2(q) SPECIAL (\n or N) -- what the pair counts of row y do
 *                not see of the symbol y (such positions exist only where a sequence starts or an N run ends: an index with 2^24 of them
 *                gets no PAIRS image); bit 31 of dw 16: the block holds a special position (c1 or c2 is \n or N);
 *                RUN CONTINUATION (round 4): bits 24..31 of dw 17 = number of positions right behind the block (at most 255) that carry the same
 *                regular PAIR as the block's last position; bits 24..31 of dw 18 = the same for the first symbol alone (any second symbol).
 *                In a pangenome of H haplotypes an interval is ~H positions wide and mostly ONE run (the haplotypes agree): where it runs on
 *                behind the block, its far end usually lies inside that run, and the counts of the part behind the block follow from the last
 *                position's pair -- no second line (H = 96 at n = 640 M: 44 % of the lane trips fetched one before).  PGX_PAIRS_EXT=0 builds
 *                the image without (fields zero).
 *   dw 20..22 / 23..25   bit planes of c1 (bit 0, bit 1);  dw 26..28 / 29..31  bit planes of c2
 * A kernel probe reads the row of its first symbol (16 bytes), dw 16..19 and the planes (48 bytes).  A kernel uses unflagged blocks
 * only, and two neighbouring blocks together only when neither is flagged: every count that involves \n or N then cancels out of the
 * differences it needs.  Anything else goes to the image it accompanies (pgx_find_mems_pairs_kernel hands such reads on).
 * (Until round 3 the dw 16 + y counts lived in an LDS table indexed by a per-block run count, which limited the image to indexes
 * with at most 1023 special runs -- a few dozen sequences; now any number of sequences qualifies.) */
#define PGX_PAIRS_SYMS 96u
#define PGX_PAIRS_BLOCK_BYTES 128u
/* PAIRS at stride 64 (PgxConsts.pairs_stride = 64; the default up to 64 GiB of image, narrow and WIDE): the
 * same 96-position blocks, but one every 64 positions -- block b covers [64 b, 64 b + 96) and its counts are those before 64 b --, so that an
 * interval of up to 32 positions lies inside ONE block wherever it starts.  At stride 96 an interval that crosses a block border costs a second
 * line (6 % of the kernel's lines at chr22 scale, where intervals are a handful of positions wide); here only intervals that end beyond
 * position 96 of their block do.  The second block of such an interval overlaps the first by 32 positions: its part starts at position 32.
 * (Round 3 also tried blocks of 64 positions read as three 16-byte pieces instead of five: fewer requests per line changed nothing, the extra
 * lines of the shorter blocks did -- profiles/r03_pairs64_layout.json -- and the layout was dropped.) */
#define PGX_PAIRS_STRIDE64 64u
/* LCE image (round 4; device only, built on the device next to a narrow PAIRS image: pgx_runtime.hip ensure_lce):
 *   lce_sa[i]     u32: where suffix i of the BWT order starts in lce_text (the r-index's samples expanded to the whole suffix array by the locate
 *                 kernels, sequence * max_length + offset -> start of the sequence + offset)
 *   lce_text      the collection itself, two bits per symbol, 16 symbols per u32, A C T G = 0 1 2 3 (the order of the packed reads: XOR compares
 *                 16 symbols at once); every sequence followed by its endmarker; N and endmarkers read as 0 and are covered by
 *   lce_flags     one bit per 128-byte line of lce_text (512 symbols): the line holds an N, an endmarker, or lies behind the text
 *   lce_lcp[i]    u8: the number of symbols suffix i has in common with suffix i - 1 (pgx_lce_lcp_kernel, from lce_text), 254 = "254 or more", 255 = not
 *                 known (the comparison touched a flagged line).  After ONE occurrence t - 1 of an interval has been compared with the text (match l), the
 *                 next one matches min(l, lcp[k + t] - symbols matched before the stage) -- unless the two are equal and the read goes on, where occurrence
 *                 t may match further and is compared itself: a forward stage is ~1.7 comparisons instead of one per occurrence (tests/test_lce_math.py).
 *                 Used while (read length - MEM start) <= 253, so that a capped entry is longer than anything asked; PGX_FM_LCP=0: not built.
 * 5.25 n bytes (chr22 scale: 3.4 GB).  The forward stage of find_mems_function (algorithm.hpp:676-700: forward_extend until the interval is "small") over
 * an interval of s <= 128 occurrences (16 without lce_lcp) is finished by comparing the read with the text at occurrences of SA[k] .. SA[k + s - 1] -- the
 * occurrences that match longest are consecutive and ARE the interval the extensions would end with -- in a few trips (one comparison, then sixteen entries of
 * lce_lcp per trip; a further comparison where two lengths tie) instead of (match length) / 2 trips of one line.  The text
 * is recovered from the index alone: the first symbol of suffix i is the symbol whose C-bucket holds i. */
/* WIDE variants of DENSE2 and PAIRS (BWTs of 2^32 symbols or more, up to PGX_SB_MAX superblocks; FastLocate is size_t end to end,
 * r-index.hpp:118-130): the same 128-byte blocks, but every count in a block header is a 32-bit DELTA against its superblock --
 * 2^sb_shift consecutive blocks, at most 2^31 symbols -- whose 64-bit bases sit in a small table the kernels stage in LDS:
 *   sbase2[8 s ..]   DENSE2, superblock s: counts of A C G T N before it, their sum, two unused words
 *   pbase[24 s ..]   PAIRS, superblock s: the sixteen pair counts before it (4 y + x), the four row sums (16 + y), four unused words
 * Positions, interval coordinates, C and pair_t2w are 64-bit in the kernels that walk these.  A narrow image is the special case of one
 * superblock with zero bases (the narrow kernels never read the tables).  PGX_MODE_IMAGE_WIDE forces the wide form on any index. */
#define PGX_SB_MAX 64u
#define PGX_D2_SB_SHIFT 22u    /* 384 * 2^22 = 1.6e9 symbols per superblock */
#define PGX_PAIRS_SB_SHIFT 24u /*  96 * 2^24 = 1.6e9 */

/* ext_tab entry (one per byte value and direction): how to extend by that byte */
#define PGX_EXT_CV(e) ((e) & 7u)            /* nuc code whose rank gives the new interval     */
#define PGX_EXT_V(e) (((e) >> 3) & 7u)      /* slot into C[]                                  */
#define PGX_EXT_M(e) (((e) >> 6) & 0x3FFFFu) /* 6 x 3-bit multiplicities over codes (k' update) */
#define PGX_EXT_KILL(e) (((e) >> 24) & 1u)  /* extension always yields the empty interval     */
#define PGX_EXT_MAKE(cv, v, m, kill) \
    ((uint32_t)(cv) | ((uint32_t)(v) << 3) | ((uint32_t)(m) << 6) | ((uint32_t)(kill) << 24))

typedef struct {
    uint64_t n;           /* bwt size (sequence_size) */
    uint64_t C[8];        /* C[slot], r-index.hpp:310 */
    uint32_t ext_tab[512]; /* [0..255] backward by byte, [256..511] forward by byte */
    uint32_t slot_code[8]; /* rank cache slot i -> nuc code (rank_at_cached_encoded view) */
    uint32_t sigma;
    uint32_t excl_mask;   /* codes whose header count is NOT part of block_start (legacy quirk 1) */
    uint32_t dir_shift;
    uint32_t n_blocks;
    uint64_t dir_entries;
    /* tags */
    uint64_t n_tag_runs;
    uint64_t tag_dir_entries;
    uint32_t tag_dir_shift;
    uint32_t has_tags;
    uint32_t mode;
    uint32_t count_supported; /* 0: COMPAT count on an encoded index without N (the reference mis-parses it, quirk 3) */
    /* unidirectional backward search (FastLocate::count / count_encoded): per read byte the nuc code
     * ranked (bits 0..2), the C slot (bits 3..5) and a "no match" flag (bit 24) */
    uint32_t cnt_tab[256];
    uint32_t image_kind;  /* PGX_IMAGE_RL / PGX_IMAGE_DENSE / PGX_IMAGE_DENSE2 */
    uint32_t has_pairs;   /* a PAIRS image accompanies the DENSE / DENSE2 image */
    uint32_t pair_t2[32]; /* [8 y + c], y = 2-bit code of a regular symbol, c = nuc code: number of c in BWT[0, first suffix starting with y) */
    uint32_t pair_runs;   /* special runs of the PAIRS image: maximal runs of positions whose c1 or c2 is \n or N (statistics only) */
    uint32_t reserved1;
    /* WIDE images (appended: the offsets above are what tests/image_emu.py reads) */
    uint32_t wide;            /* header counts are deltas against superblock bases; 64-bit kernels */
    uint32_t d2_sb_shift, pairs_sb_shift; /* blocks per superblock = 1 << shift */
    uint32_t n_sb2, n_sbp;    /* superblocks of the DENSE2 / PAIRS image */
    uint32_t pairs_stride;    /* positions between the starts of consecutive PAIRS blocks: PGX_PAIRS_SYMS (96) or PGX_PAIRS_STRIDE64 (overlapping blocks) */
    uint64_t pair_t2w[32];    /* pair_t2 in 64 bits (always filled) */
} PgxConsts;

/* locate image (FastLocate::locate / locateNext / decompressSA, src/r-index.cpp:1252-1366): three sorted
 * u64 arrays with the same sampled u32 directory as tdir (dir[i] = #elements <= i << shift):
 *   rstart[r + 1]  first BWT position of every run in the reference's run numbering (rstart[r] = n); rsamp[r] =
 *                  samples[run] = packed text position (seq * max_length + offset) of the run's first suffix
 *   lpos[r]        ones of `last` (packed text positions of run tails); lnext[i] = samples[last_to_run[i] + 1],
 *                  so locateNext(v) = lnext[i] + (v - lpos[i]) with i = predecessor of v in lpos               */
#define PGX_NO_POSITION (~(uint64_t)0)
typedef struct {
    uint64_t n;          /* bwt size */
    uint64_t n_runs;     /* samples.size() */
    uint64_t n_last;     /* ones of last */
    uint64_t max_length; /* header.max_length: seqId = v / max_length */
    uint64_t rdir_entries, ldir_entries;
    uint32_t rdir_shift, ldir_shift;
} PgxLocConsts;

#endif

/*
 * pgx_oracle.h -- CPU ORACLE for the find_mems hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a dependency-free C restatement of the reference's integer arithmetic
 * (parsaeskandar/pangenome-index @ 2025-10-24).  It may be imported / linked / executed only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and only as the checker.
 * The product (pangenome-index_amd/) never includes or links this file.
 *
 * Parity pinning: the reference cannot be built in this environment (SDSL/GBWT/GBWTGraph/grlBWT
 * are absent), so this restatement is pinned by
 *   (1) byte-exact parsing of the reference's own fixtures (xy.ri, xy_bidirectional_compressed.tags),
 *   (2) the known answers of SURVEY.md section 8c (tests/test_oracle.py),
 *   (3) brute-force substring-count truth in ORC_MODE_STRICT.
 *   (4) the reference's own Locate_* tests (tests/test_rindex.cpp:103-244) for the locate functions: decompressDA must
 *       equal the document array of a brute-force BWT (tests/test_locate.py).
 * The reference's own tests hold no golden vector for find_all_mems or the tag queries and no reference binary can be
 * built here: for those two functions this oracle is "parity unpinned" (restatement-derived; DESIGN.md says the same).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef PGX_ORACLE_H
#define PGX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MODE_COMPAT 0 /* literal reference behaviour, quirks included (SURVEY 8a quirk list) */
#define ORC_MODE_STRICT 1 /* textbook FMD extension over the true ranks (our extension)          */

#define ORC_TAGS_BYTECODE 1 /* src/tag_arrays.cpp:739-763 + query :780-854 */
#define ORC_TAGS_COMPACT 2  /* src/tag_arrays.cpp:766-776 + query :856-890 */

/* include/pangenome_index/algorithm.hpp:644-651 */
typedef struct {
    uint64_t start, end, bwt_start;
    int64_t size;
} orc_mem;

/* include/pangenome_index/r-index.hpp:118-130 */
typedef struct {
    uint64_t forward, reverse;
    int64_t size;
} orc_biint;

typedef struct orc_ri orc_ri;
typedef struct orc_tags orc_tags;

const char *orc_last_error(void);

/* ---- r-index (FastLocate::load_encoded, src/r-index.cpp:406-459; legacy fall-back :378-404) ---- */
orc_ri *orc_ri_load(const char *path);
void orc_ri_free(orc_ri *r);
uint64_t orc_ri_bwt_size(const orc_ri *r);    /* sequence_size */
uint64_t orc_ri_sigma(const orc_ri *r);       /* C.size() */
int orc_ri_is_encoded(const orc_ri *r);       /* r-index.hpp:409 */
int orc_ri_has_N(const orc_ri *r);
uint64_t orc_ri_n_blocks(const orc_ri *r);    /* blocks.size() or blocks_encoded_start_bits.size() */
uint64_t orc_ri_n_block_starts(const orc_ri *r); /* ones in blocks_start_pos */
uint64_t orc_ri_C(const orc_ri *r, uint64_t i);
uint8_t orc_ri_sym_map(const orc_ri *r, uint64_t c);
uint64_t orc_ri_block_start(const orc_ri *r, uint64_t i); /* i-th one of blocks_start_pos */
uint64_t orc_ri_max_length(const orc_ri *r);
uint64_t orc_ri_samples_size(const orc_ri *r);
uint64_t orc_ri_sample(const orc_ri *r, uint64_t i);
uint64_t orc_ri_last_ones(const orc_ri *r);
uint64_t orc_ri_last_size(const orc_ri *r);
uint64_t orc_ri_last_select(const orc_ri *r, uint64_t i); /* 0-based i-th one */
uint64_t orc_ri_last_to_run(const orc_ri *r, uint64_t i);
uint64_t orc_ri_encoded_stream_bytes(const orc_ri *r);
uint64_t orc_ri_file_bytes_consumed(const orc_ri *r);
/* legacy block access (for construction cross-checks) */
uint64_t orc_ri_block_nruns(const orc_ri *r, uint64_t b);
void orc_ri_block_run(const orc_ri *r, uint64_t b, uint64_t i, uint64_t *sym, uint64_t *len);
uint64_t orc_ri_block_cum(const orc_ri *r, uint64_t b, uint64_t i);

/* rank_at_cached_encoded (src/r-index.cpp:619-641) / rank_at_cached (:593-603): out has sigma entries */
void orc_rank_at_cached(const orc_ri *r, uint64_t pos, uint64_t *out);
/* true ranks of the six nuc codes (\n A C G N T) in BWT[0,pos) */
void orc_rank6_true(const orc_ri *r, uint64_t pos, uint64_t out[6]);

/* backward_extend_encoded (src/r-index.cpp:713-756) / forward_extend_encoded (:758-764) */
orc_biint orc_backward_extend(const orc_ri *r, int mode, orc_biint in, uint8_t a);
orc_biint orc_forward_extend(const orc_ri *r, int mode, orc_biint in, uint8_t a);

/* find_all_mems (include/pangenome_index/algorithm.hpp:739-757).  Returns number of MEMs; writes
 * at most cap of them.  n_ext (optional) accumulates the number of extensions performed. */
uint64_t orc_find_all_mems(const orc_ri *r, int mode, const uint8_t *read, uint64_t len,
                           uint64_t min_len, uint64_t min_occ, orc_mem *out, uint64_t cap,
                           uint64_t *n_ext);

/* one call of find_mems_function (algorithm.hpp:653-736) at start position x (x <= len): returns the start position the
 * function returns, the MEM it pushed (if *has_mem) and adds its extensions to *n_ext (optional) */
uint64_t orc_find_mems_function(const orc_ri *r, int mode, const uint8_t *read, uint64_t len, uint64_t min_len, uint64_t min_occ,
                                uint64_t x, orc_mem *mem, int *has_mem, uint64_t *n_ext);

/* ---- query_tags path: FastLocate::count / count_encoded, r-index.hpp:540-556 ---- */
/* returns the final range; an empty range is (1, 0) like the reference's {1, 0}.
 * COMPAT: literal LF (src/r-index.cpp:650-687, legacy: unknown symbols are rejected :653) or LF_encoded
 * (:689-711 over rankAt_encoded :570-590, which always reads SIX cumulative varints: quirk 3 --
 * wrong-but-deterministic on an encoded index without N).  STRICT: textbook backward search. */
void orc_count(const orc_ri *r, int mode, const uint8_t *read, uint64_t len, uint64_t *first, uint64_t *second);
/* one step of it: FastLocate::LF (src/r-index.cpp:650-687) / LF_encoded (:689-711) on the inclusive range [*first, *second] */
void orc_LF(const orc_ri *r, int mode, uint8_t sym, uint64_t *first, uint64_t *second);

/* ---- locate: src/r-index.cpp:1252-1366, r-index.hpp:424-436,490-501 ---- */
#define ORC_NO_POSITION (~(uint64_t)0)
uint64_t orc_locate_first(const orc_ri *r);
uint64_t orc_locate_next(const orc_ri *r, uint64_t prev);
uint64_t orc_seq_id(const orc_ri *r, uint64_t v);
uint64_t orc_seq_offset(const orc_ri *r, uint64_t v);
void orc_decompress_sa(const orc_ri *r, uint64_t *out); /* out has bwt_size entries; DA = seq_id of each */
/* SA values of BWT[first..last] in BWT order (what locate / locate_encoded compute before seqId + sort + unique);
 * returns their number, or ORC_NO_POSITION when the reference's literal scan is undefined (COMPAT, encoded, no N) */
uint64_t orc_locate_sa(const orc_ri *r, int mode, uint64_t first, uint64_t last, uint64_t *out);

/* ---- tag array ---- */
orc_tags *orc_tags_load(const char *path, int format);
void orc_tags_free(orc_tags *t);
uint64_t orc_tags_n_runs(const orc_tags *t);       /* ones in bwt_intervals */
uint64_t orc_tags_bwt_intervals_size(const orc_tags *t);
uint64_t orc_tags_n_items(const orc_tags *t);      /* decoded items in the run stream */
uint64_t orc_tags_n_starts(const orc_tags *t);
uint64_t orc_tags_start(const orc_tags *t, uint64_t i);    /* i-th one of encoded_runs_starts_sd */
uint64_t orc_tags_interval(const orc_tags *t, uint64_t i); /* i-th one of bwt_intervals */
uint64_t orc_tags_item(const orc_tags *t, uint64_t i);     /* raw item value */
uint64_t orc_tags_file_bytes_consumed(const orc_tags *t);
/* query_compressed / query_compressed_compact: returns #unique positions (sorted into out, at most
 * cap written), *run_nums = number_of_runs, *overflow = 1 when the reference would read past the
 * stored runs (undefined behaviour there; defined here as value 0). */
uint64_t orc_tags_query(const orc_tags *t, uint64_t start, uint64_t end, uint64_t *run_nums,
                        uint64_t *out, uint64_t cap, int *overflow);

/* ---- whole-batch driver = the CPU baseline (find_mems.cpp:94-139 without the printing) ---- */
typedef struct {
    uint64_t n_reads;
    uint64_t *mem_offsets; /* n_reads+1 */
    orc_mem *mems;
    uint64_t *tag_run_counts; /* per MEM (NULL without tags) */
    uint64_t *pos_offsets;    /* n_mems+1 (NULL without tags) */
    uint64_t *positions;
    uint64_t n_extensions;
    uint64_t n_tag_overflow;
    double seconds_mems, seconds_tags;
} orc_batch_result;

/* reads: concatenated bytes, read i = [offsets[i], offsets[i+1]).  threads<=1: sequential (the
 * reference's execution model); otherwise OpenMP parallel for schedule(dynamic,256) over reads. */
orc_batch_result *orc_find_mems_batch(const orc_ri *r, const orc_tags *t, int mode,
                                      const uint8_t *reads, const uint64_t *offsets,
                                      uint64_t n_reads, uint64_t min_len, uint64_t min_occ,
                                      int threads);
void orc_batch_free(orc_batch_result *res);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif

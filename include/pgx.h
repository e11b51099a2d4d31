/*
 * pgx.h -- C ABI of the MI355X-native find_mems path (libpgx.so).
 *
 * The reference (parsaeskandar/pangenome-index @ 2025-10-24) has no FFI layer: its boundary is the
 * `find_mems` process contract plus C++ member functions compiled into each CLI.  This header is
 * the flat boundary a maintainer would bind instead; every entry point names the reference code
 * it replaces (paths relative to /root/reference).  Plain pointers and sizes only; no exceptions
 * cross this boundary; every function returns a pgx_status and pgx_last_error() holds the text.
 *
 * There is NO CPU fallback behind these entry points: anything that computes (rank / extend /
 * find_mems / tag queries) runs as HIP kernels on a gfx950 device and fails with
 * PGX_ERR_NO_DEVICE when none is usable.
 */
#ifndef PGX_H
#define PGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGX_ABI_VERSION 4 /* 2: pgx_timing grew (pairs_reads, redo_reads), pgx_index_info.image_pairs, PGX_MODE_IMAGE_PAIRS; 3: pgx_timing grew (ms_find_mems_main, traffic counters); 4: pgx_timing.ms_per_upload, pgx_pack_reads, pgx_batch_upload_packed */

typedef enum {
    PGX_OK = 0,
    PGX_ERR_IO = 1,          /* cannot open / read (find_mems.cpp:30,91 exit(EXIT_FAILURE)) */
    PGX_ERR_FORMAT = 2,      /* sdsl::simple_sds::InvalidData in the reference (r-index.cpp:412-419) */
    PGX_ERR_UNSUPPORTED = 3, /* valid file, but a shape this build does not handle */
    PGX_ERR_NO_DEVICE = 4,   /* no usable HIP device / kernel image */
    PGX_ERR_HIP = 5,         /* a HIP runtime call failed */
    PGX_ERR_ARG = 6,
    PGX_ERR_NOMEM = 7
} pgx_status;

/* mode: which extension tables are loaded into the device image (same kernels either way) */
#define PGX_MODE_COMPAT 0u /* bit-exact with the reference, quirks included (default)          */
#define PGX_MODE_STRICT 1u /* textbook FMD over the true ranks; equals COMPAT on sigma=6 indexes */
/* optional bits or-ed into `mode`: force the layout of the device rank image (default: chosen from the index size;
 * results never depend on it).  PGX_ERR_UNSUPPORTED if dense is forced on a legacy-layout index without N in COMPAT. */
#define PGX_MODE_IMAGE_RL 0x100u    /* run-length blocks + directory (any size)                     */
#define PGX_MODE_IMAGE_DENSE 0x200u /* uncompressed bit planes, n bytes of device memory (64 symbols per 64-byte block)       */
#define PGX_MODE_IMAGE_DENSE2 0x400u /* two bit planes + exception runs, n / 3 bytes (384 symbols per 128-byte block; n < 2^32) */
#define PGX_MODE_IMAGE_PAIRS 0x800u  /* dense2 + the two-step PAIRS image, 4 n / 3 bytes more (n < 2^32, textbook extension tables, few N runs);
                                      * by default a PAIRS image accompanies whichever dense layout is chosen, when the index qualifies */
#define PGX_MODE_IMAGE_WIDE 0x1000u  /* modifier: the 64-bit form of dense2 / PAIRS (header counts as deltas against superblock bases, 64-bit
                                      * kernels), automatic for BWTs of 2^32 symbols or more; alone it selects dense2 + PAIRS */
#define PGX_MODE_MASK 0xFFu

/* tag file formats (SURVEY section 5 "Tag formats") */
#define PGX_TAGS_AUTO 0u
#define PGX_TAGS_BYTECODE 1u /* TagArray::load_compressed_tags      src/tag_arrays.cpp:739-763 */
#define PGX_TAGS_COMPACT 2u  /* TagArray::load_compressed_tags_sdsl src/tag_arrays.cpp:766-776 */

/* struct MEM, include/pangenome_index/algorithm.hpp:644-651 (32 bytes, same field order) */
typedef struct {
    uint64_t start;
    uint64_t end;
    uint64_t bwt_start;
    int64_t size;
} pgx_mem;

/* FastLocate::bi_interval, include/pangenome_index/r-index.hpp:118-130 */
typedef struct {
    uint64_t forward;
    uint64_t reverse;
    int64_t size;
} pgx_biint;

typedef struct pgx_index pgx_index; /* FastLocate + TagArray, host image + per-device images */
typedef struct pgx_batch pgx_batch; /* one device-resident read batch and its results */

typedef struct {
    uint64_t bwt_size;      /* FastLocate::bwt_size(), r-index.hpp:568 */
    uint64_t sigma;         /* C.size() */
    uint64_t n_sequences;   /* FastLocate::tot_strings(), r-index.hpp:484 */
    uint64_t n_ref_blocks;  /* blocks in the file (10 runs each, r-index.hpp:312) */
    uint64_t n_runs;        /* logical BWT runs held by the image */
    uint64_t n_dev_blocks;  /* 64-byte device rank blocks */
    uint64_t dir_entries;   /* directory entries (u64) */
    uint32_t dir_shift;
    uint32_t is_encoded;    /* FastLocate::is_encoded(), r-index.hpp:409 */
    uint32_t has_N;         /* encoded_has_N */
    uint32_t mode;
    uint32_t has_tags;
    uint32_t tag_format;
    uint64_t n_tag_runs;    /* ones in bwt_intervals */
    uint64_t tag_dir_entries;
    uint32_t tag_dir_shift;
    uint32_t image_in_lds;  /* 1 when the rank image fits the per-workgroup LDS budget */
    uint64_t image_bytes;   /* device bytes of the rank image (blocks + directory + block lows) */
    uint64_t tag_image_bytes;
    double ref_block_mean_bytes; /* mean encoded block size of the reference layout (B_blk) */
    uint64_t max_length;    /* Header::max_length: packed position = seq * max_length + offset (r-index.hpp:424) */
    uint64_t n_samples;     /* samples.size() = runs in the reference's numbering */
    uint32_t image_kind;    /* 0 = run-length blocks + directory, 1 = dense bit planes (64 symbols per 64-byte block),
                             * 2 = dense2 (384 symbols per 128-byte block, two planes + exception runs) */
    uint32_t image_pairs;   /* 1 = a PAIRS image (two extensions per cache line) accompanies the dense / dense2 image */
    uint32_t image_wide;    /* 1 = the 64-bit form of dense2 / PAIRS (BWTs of 2^32 symbols or more, or PGX_MODE_IMAGE_WIDE) */
    uint32_t pairs_stride;  /* positions between the starts of consecutive blocks of the PAIRS image (each covers 96): 96, or 64 = overlapping blocks; 0 without one */
} pgx_index_info;

const char *pgx_last_error(void);
int pgx_abi_version(void);

/* ---- index lifetime ---------------------------------------------------------------------- */
/* Replaces FastLocate::load_encoded (src/r-index.cpp:406-459, legacy fall-back :378-404) and
 * TagArray::load_compressed_tags{,_sdsl} (src/tag_arrays.cpp:739-776).  tags_path may be NULL.
 * Host-only: parses the files and builds the flat device image in host memory. */
pgx_status pgx_index_open(const char *ri_path, const char *tags_path, uint32_t tags_format,
                          uint32_t mode, pgx_index **out);
/* Same from memory images of the files (what std::istream-based loaders hand over).  Either image
 * may be NULL: a tags-only handle serves TagArray, an r-index-only handle serves FastLocate. */
pgx_status pgx_index_open_memory(const void *ri_bytes, uint64_t ri_n, const void *tags_bytes, uint64_t tags_n,
                                 uint32_t tags_format, uint32_t mode, pgx_index **out);
/* FastLocate's public tables: sym_map (r-index.hpp:307), C (:310), complement_table (:347) */
pgx_status pgx_index_tables(const pgx_index *h, uint8_t sym_map[256], uint64_t C[8], uint8_t complement[256]);
pgx_status pgx_index_info_get(const pgx_index *h, pgx_index_info *info);
/* Frees the host image and every device image.  Batches created from the index use its device images: free them first. */
void pgx_index_close(pgx_index *h);

/* Copy the image to a HIP device (idempotent per device). */
pgx_status pgx_index_to_device(pgx_index *h, int device);

/* Host views of the flat image (for tests that verify the layout without a GPU, and FastLocate::getSample).  `which`:
 * 0 rank blocks (64 B each), 1 directory (u64), 2 block starts (u64, host only), 3 tag run starts (u64),
 * 4 tag values (u64), 5 tag directory (u32), 6 constants table (PgxConsts, pgx_image.h), 7 block lows (u16);
 * locate image (built on first use of one of these selectors, pgx_image.h "locate image"):
 * 8 rstart (u64, first BWT position of every run + n), 9 rsamp (u64, samples[run]: what getSample returns),
 * 10 rdir (u32), 11 lpos (u64, ones of `last`), 12 lnext (u64, samples[last_to_run[i] + 1]), 13 ldir (u32),
 * 14 locate constants (PgxLocConsts); 15 exception runs of the dense2 rank image (u32);
 * literal count image of an encoded index without N: 16 block starts (u64), 17 six cumulative counts per
 * block (u64), 18 runs as the reference's late scan sees them (u64: code << 56 | length), 19 first run of every block (u32);
 * two-step PAIRS image (pgx_image.h; empty without one): 20 blocks (32 dwords each);
 * WIDE images: 22 superblock bases of the dense2 image (8 u64 each), 23 of the PAIRS image (24 u64 each). */
pgx_status pgx_index_image_view(const pgx_index *h, int which, const void **ptr, uint64_t *bytes);

/* The DEVICE copy of a view (tests: what the kernels read must be what the host built): which = 0 (rank blocks), 15 (exception runs),
 * 20 (PAIRS blocks), 22 / 23 (superblock bases); copies min(bytes, size of the view) bytes into out.
 * 30 .. 33: the LCE image, which exists on the device only (pgx_image.h; built by this call if it has not been): suffix array in text
 * coordinates (u32 x n), text at two bits per symbol, flag bits per 128-byte text line, common prefixes of neighbouring suffixes (u8 x n);
 * nothing is copied where the index has no such image. */
pgx_status pgx_index_device_view(pgx_index *h, int device, int which, void *out, uint64_t bytes);

/* ---- index construction (build side; CPU, run once) --------------------------------------- */
/* Replaces build_rindex (src/build_rindex.cpp:13-21 -> FastLocate(std::string) src/r-index.cpp:778
 * + serialize_encoded :297-376).  encoded=0 writes the legacy layout (serialize, :266-294). */
pgx_status pgx_build_rindex(const char *rlbwt_path, const char *out_ri_path, int encoded);
/* BWT of a newline-terminated sequence collection (what grlBWT produces for the reference):
 * writes the grlBWT .rl_bwt layout read by bwt_buff_reader. */
pgx_status pgx_build_rlbwt(const char *text_path, const char *out_rlbwt_path);
/* Both steps in one call (out_rlbwt_path may be NULL): the suffix array the BWT is made from also yields the SA samples
 * directly, so the sampling walk of FastLocate(std::string) (src/r-index.cpp:993-1130) is not repeated; the .ri written
 * is byte-identical to pgx_build_rlbwt + pgx_build_rindex. */
pgx_status pgx_build_index_from_text(const char *text_path, const char *out_rlbwt_path, const char *out_ri_path, int encoded);
/* The same for a collection handed over as several texts ("chromosomes"; their sequences in the order of the texts): one suffix array
 * per text, built side by side, then a k-way merge of the sorted suffix lists split over host threads.  The files are byte-identical
 * to what the single call writes for the concatenation; every text must stay below 2^31 symbols, the collection may have any size
 * (the reference takes grlBWT's output of any size: FastLocate(std::string), src/r-index.cpp:778-1139). */
pgx_status pgx_build_index_from_texts(const char *const *text_paths, uint32_t n_texts, const char *out_rlbwt_path, const char *out_ri_path, int encoded);
/* Write a compact sdsl tag file (format 3, src/tag_arrays.cpp:940-974 + :622-654) from parallel
 * arrays of run values (already `offset | rev<<10 | node<<11`) and run lengths. */
pgx_status pgx_write_compact_tags(const char *out_path, const uint64_t *values,
                                  const uint64_t *lengths, uint64_t n_runs);

/* convert_tags equivalent (src/convert_tags.cpp): build_tags' "algorithm format" -> a query format.
 * compact = 0: ByteCode format (load_compressed_tags, tag_arrays.cpp:739); 1: sdsl-compact (find_mems.cpp:79). */
pgx_status pgx_convert_tags(const char *in_path, const char *out_path, int compact);

/* merge_tags equivalent (src/merge_tags.cpp): per-chromosome tag streams (build_tags' "algorithm format": ByteCode runs
 * offset:10 | rev:1 | len:9 | node<<20, with or without the 8-byte int_vector<8> header of sdsl::int_vector_buffer) ->
 * the whole-genome tag array in the sdsl-compact query format find_mems loads.  `ri_path` is the whole-genome r-index;
 * seq_to_file[s] (n_seq = tot_strings entries) names the tag file of sequence s -- the reference derives this from the
 * GBZ (first node of path s -> weakly connected component -> the file whose first tag lies in that component,
 * merge_tags.cpp:478-512); a GBZ reader is out of scope here, so the caller states it.  Runs on `device`: document
 * array by the locate kernels, one scan + gather per file, run-length encoding.  Merged runs are maximal (the reference
 * counts them in a uint16_t, which wraps beyond 65 535) and split at 511 like append_compact_run_streamed
 * (src/tag_arrays.cpp:940-974).  PGX_ERR_FORMAT when a file holds a different number of tags than the BWT has
 * positions of its sequences. */
pgx_status pgx_merge_tags(const char *ri_path, const char *const *tag_paths, uint32_t n_files, const uint32_t *seq_to_file,
                          uint64_t n_seq, int device, const char *out_path);
/* As pgx_merge_tags, with flags.  PGX_MERGE_REFERENCE_RUNS: the file byte for byte as the reference writes it.  The reference
 * collects runs job by job (500 BWT runs each, src/merge_tags.cpp:600,733-823) in std::pair<pos_t, uint16_t> and joins the last
 * run of one job to the first of the next when the tags are equal (:776-777, previous_last_run), so what it hands to
 * append_compact_run_streamed is every MAXIMAL run with its length taken mod 65 536 -- a length that lands on 0 writes nothing
 * (src/tag_arrays.cpp:959) and the positions of every later run shift down.  Without the flag (and in pgx_merge_tags) lengths are
 * exact; the two outputs are the same bytes whenever no merged run reaches 65 536 positions (tests/test_merge_tags.py restates
 * the reference's job loop and checks that rule). */
#define PGX_MERGE_REFERENCE_RUNS 0x1u
pgx_status pgx_merge_tags_ex(const char *ri_path, const char *const *tag_paths, uint32_t n_files, const uint32_t *seq_to_file,
                             uint64_t n_seq, int device, const char *out_path, uint32_t flags);

/* What merge_tags asks the graph (src/merge_tags.cpp:443-445,478-515; include/pangenome_index/algorithm.hpp:600-619): for every
 * GBWT sequence of the GBZ the graph node id of its first node (gbz.index.extract(i)[0]; 0 for an empty path) and that node's
 * weakly connected component (numbered by smallest node id), plus the graph's largest node id and component count.  Reads only
 * the GBWT's node records of the file (simple-sds layout).  first_node / component may be NULL; at most cap entries are written. */
pgx_status pgx_gbz_paths(const char *gbz_path, uint64_t *n_sequences, uint64_t *first_node, uint32_t *component, uint64_t cap,
                         uint64_t *max_node_id, uint32_t *n_components);
/* merge_tags with the reference's own inputs: graph.gbz, whole-genome r-index, per-chromosome tag files.  Sequence s belongs
 * to the tag file whose first tag lies in the component of the first node of path s (merge_tags.cpp:478-515); the item width
 * of the output comes from the graph's largest node id (:627-638).  Otherwise as pgx_merge_tags. */
pgx_status pgx_merge_tags_gbz(const char *gbz_path, const char *ri_path, const char *const *tag_paths, uint32_t n_files, int device,
                              const char *out_path);
pgx_status pgx_merge_tags_gbz_ex(const char *gbz_path, const char *ri_path, const char *const *tag_paths, uint32_t n_files, int device,
                                 const char *out_path, uint32_t flags); /* flags as pgx_merge_tags_ex */

/* ---- primitives (tests; mirror the public FastLocate / TagArray query API) ----------------- */
/* FastLocate::rank_at_cached_encoded (src/r-index.cpp:619-641): out[i*6 .. i*6+sigma) per position;
 * entries >= sigma are zero.  true_codes!=0 returns the six true code ranks instead. */
pgx_status pgx_rank_batch(pgx_index *h, int device, const uint64_t *pos, uint64_t n,
                          int true_codes, uint64_t *out);
/* FastLocate::backward_extend_encoded / forward_extend_encoded (src/r-index.cpp:713-764) */
pgx_status pgx_extend_batch(pgx_index *h, int device, const pgx_biint *in, const uint8_t *sym,
                            const uint8_t *forward, uint64_t n, pgx_biint *out);
/* TagArray::query_compressed{,_compact} (src/tag_arrays.cpp:780-890) without the printing.
 * Two-call pattern: positions==NULL returns counts only.  pos_offsets has n+1 entries. */
pgx_status pgx_tag_query_batch(pgx_index *h, int device, const uint64_t *start, const uint64_t *end,
                               uint64_t n, uint64_t *run_nums, uint64_t *pos_offsets,
                               uint64_t *positions, uint64_t positions_cap, uint64_t *n_overflow);

/* FastLocate::count / count_encoded (include/pangenome_index/r-index.hpp:540-556): backward search of
 * every read; out[i] = final BWT range, {1, 0} when empty (the query_tags path, src/query_tags.cpp:88-96).
 * In COMPAT mode on an encoded index without N the reference's rankAt_encoded mis-parses every block (it always reads
 * six cumulative varints, src/r-index.cpp:578: SURVEY 8a quirk 3); that wrong-but-deterministic result is reproduced
 * (a separate image of the reference's blocks as its scan sees them); PGX_MODE_STRICT gives the true ranges. */
typedef struct {
    uint64_t first;
    uint64_t second;
} pgx_range;
pgx_status pgx_count_batch(pgx_index *h, int device, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads,
                           pgx_range *out);

/* One step of it -- FastLocate::LF (src/r-index.cpp:650-687) / LF_encoded (:689-711): out[i] = LF(in[i], sym[i]) for n
 * inclusive ranges; empty inputs and results are {1, 0}.  Same COMPAT restriction as pgx_count_batch. */
pgx_status pgx_lf_batch(pgx_index *h, int device, const pgx_range *in, const uint8_t *sym, uint64_t n, pgx_range *out);

/* ---- locate: suffix-array samples of the r-index ---------------------------------------- */
#define PGX_NO_POSITION (~(uint64_t)0)
#define PGX_LOCATE_SEQ_IDS 1u /* values are seqId(v) = v / max_length (r-index.hpp:429) instead of packed positions */
#define PGX_LOCATE_UNIQUE 2u  /* every query's values sorted and de-duplicated (src/r-index.cpp:1293-1294)            */
/* FastLocate::locate / locate_encoded (src/r-index.cpp:1252-1341) for n BWT ranges [first[i], last[i]] (inclusive;
 * last < first is the empty state).  flags = PGX_LOCATE_SEQ_IDS | PGX_LOCATE_UNIQUE is the reference's result
 * (sorted unique sequence ids); flags = 0 gives the packed suffix-array values pack(seq, offset) in BWT order.
 * val_offsets has n + 1 entries.  values == NULL returns the offsets only; otherwise values_cap must hold the
 * result (sum of last - first + 1 always suffices).  Ranges must end before bwt_size (PGX_ERR_ARG otherwise).
 * PGX_ERR_UNSUPPORTED in COMPAT mode on an encoded index without N: the reference's run scan skips six header
 * varints where five were written (EncodedBlock::skip_header, src/r-index.cpp:83-88). */
pgx_status pgx_locate_batch(pgx_index *h, int device, const uint64_t *first, const uint64_t *last, uint64_t n,
                            uint32_t flags, uint64_t *val_offsets, uint64_t *values, uint64_t values_cap);
/* FastLocate::locateNext (src/r-index.cpp:1363-1366) for n packed positions; PGX_NO_POSITION where the reference's
 * result is undefined (no tail sample at or before prev, or the tail of the last run). */
pgx_status pgx_locate_next_batch(pgx_index *h, int device, const uint64_t *prev, uint64_t n, uint64_t *out);
/* FastLocate::decompressSA (flags = 0) / decompressDA (flags = PGX_LOCATE_SEQ_IDS), src/r-index.cpp:1343-1361:
 * out has bwt_size entries.  Every BWT run is an independent chain from its head sample. */
pgx_status pgx_decompress_sa(pgx_index *h, int device, uint32_t flags, uint64_t *out);

/* ---- the hot path: find_mems over a batch of reads ----------------------------------------- */
#define PGX_RUN_TAGS 1u    /* also run the tag queries of find_mems.cpp:129 */
#define PGX_RUN_TIMING 2u  /* record HIP events around each kernel (pgx_batch_timing)           */

typedef struct {
    uint64_t n_reads;
    uint64_t n_mems;
    const uint64_t *mem_offsets;    /* n_reads+1; MEMs of read i = mems[mem_offsets[i] .. [i+1])   */
    const pgx_mem *mems;            /* discovery order (find_all_mems, algorithm.hpp:739-757)     */
    const uint64_t *tag_run_counts; /* per MEM: number_of_runs (tag_arrays.cpp:860); NULL w/o tags */
    const uint64_t *pos_offsets;    /* n_mems+1; NULL w/o tags                                    */
    const uint64_t *positions;      /* sorted unique graph positions per MEM (tag_arrays.cpp:882) */
    uint64_t n_positions;
    uint64_t n_extensions;          /* backward+forward extensions performed (counter)           */
    uint64_t n_tag_overflow;        /* tag queries that read past the stored runs (ref: UB)      */
} pgx_result;

typedef struct {
    float ms_find_mems;   /* the dominant kernel (pgx_find_mems_kernel) */
    float ms_compact;     /* MEM compaction + scans */
    float ms_tag_locate;
    float ms_tag_gather;
    float ms_tag_sort;
    float ms_total;       /* first launch -> last launch, device time */
    uint32_t find_mems_launches;
    uint32_t heavy_reads; /* reads whose rest went through the heavy-read kernel (filled by every run, timed or not) */
    uint32_t pairs_reads; /* != 0: the run used the two-step PAIRS kernel (2: with the reads packed in LDS; 3: and cooperative line fetches) */
    uint32_t pairs_other_steps; /* extensions the PAIRS kernel took through the image it accompanies: its own block held \n or N, or the interval was wider
                                 * than two blocks (until ABI 3's last revision: redo_reads, reads handed on to the dense2 kernel) */
    /* the first launch of the find_mems stage alone (the PAIRS kernel when pairs_reads, else pgx_find_mems_kernel): ms_find_mems
     * also covers the launches behind it (reads handed on, heavy reads) */
    float ms_find_mems_main;
    uint32_t seed_depth;  /* depth K of the k-mer seed table the run used (0 = none: no table, or min_len below every table's depth) */
    /* what the kernels asked of the memory system, counted by the kernels themselves (filled by every run, timed or not; zero for
     * images staged in LDS): 128-byte lines of the rank image fetched by rank probes -- trips whose line every lane shares (a
     * stage's first probe of the full interval) are not counted -- and seed / end table entries read (16 bytes, one line each) */
    uint64_t main_lines, main_seed_loads;   /* the launch ms_find_mems_main times */
    uint64_t other_lines, other_seed_loads; /* the other pgx_find_mems_kernel launches of the run */
    uint64_t two_step_trips;                /* PAIRS kernel: lane trips that performed two extensions from one line */
    /* Device time every FRESH batch pays before its first find_mems launch and a re-run of resident reads does not: with pgx_batch_upload the
     * pass that finds the chunks holding a byte outside A C G T and packs the reads to two bits per symbol, its read-back, the pass that lists
     * the reads concerned, and the scan of the worst-case MEM slots -- all inside the first pgx_batch_run after the upload (from its first
     * event to the first find_mems launch); with pgx_batch_upload_packed the unpack pass and the listed reads' bytes (inside the upload) plus
     * that scan.  0 for a run that found everything in place. */
    float ms_per_upload;
} pgx_timing;

/* Upload reads (read i = reads[offsets[i] .. offsets[i+1]); the `std::getline` lines of
 * find_mems.cpp:96-98, empty lines already skipped by the caller) to `device`. */
pgx_status pgx_batch_create(pgx_index *h, int device, const uint8_t *reads, const uint64_t *offsets,
                            uint64_t n_reads, pgx_batch **out);
/* Pinned host memory for the read bytes handed to pgx_batch_create / pgx_batch_upload (and for anything else the caller streams to a
 * device): uploads from it run at the speed of the link, uploads from ordinary pageable memory are staged by the runtime at a fraction of
 * it (find_mems CLI, 16 M reads: 0.95 s of 1.9 s pipeline wall went into pageable uploads).  PGX_ERR_NO_DEVICE without a GPU. */
pgx_status pgx_host_alloc(size_t bytes, void **out);
void pgx_host_free(void *p);
/* Replace the reads of an existing batch (its device and pinned host buffers only ever grow: a long-lived
 * batch costs no allocation per call).  Invalidates the results of the previous run. */
pgx_status pgx_batch_upload(pgx_batch *b, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads);
/* The same with the reads packed by the caller (the find_mems CLI's parse threads do it): a quarter of the bytes over the link, and no pass over
 * the read bytes on the device before the search.  `packed`: two bits per symbol, (byte >> 1) & 3 (A C T G = 0 1 2 3), symbol i of the
 * concatenation reads[offsets[0] ..) in bits 2 (i & 15) of word i >> 4, (offsets[n_reads] + 15) / 16 words; offsets[0] must be 0.  Reads that hold
 * any other byte are listed -- side_ids ascending, their bytes as they are concatenated in side_bytes in list order -- and what the packed words
 * say about them is ignored.  pgx_pack_reads produces all of it from a batch in the form pgx_batch_upload takes; results are the same bytes
 * either way (tests/test_gpu_parity.py).  Replaces the same per-read loop: src/find_mems.cpp:94-139. */
pgx_status pgx_batch_upload_packed(pgx_batch *b, const uint32_t *packed, const uint64_t *offsets, uint64_t n_reads, const uint64_t *side_ids,
                                   const uint8_t *side_bytes, uint64_t n_side);
/* Host side of it (no device needed): packs reads[offsets[0] .. offsets[n_reads]) into `packed` ((bytes + 15) / 16 words) on `threads` host
 * threads (0 = half the cores, at most 16) and lists the reads with a byte outside A C G T (upper case) with their bytes.  *n_side / *n_side_bytes
 * are always what the batch needs; PGX_ERR_NOMEM when that exceeds side_ids_cap / side_bytes_cap (upload such a batch as bytes). */
pgx_status pgx_pack_reads(const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads, uint32_t threads, uint32_t *packed, uint64_t *side_ids,
                          uint64_t side_ids_cap, uint8_t *side_bytes, uint64_t side_bytes_cap, uint64_t *n_side, uint64_t *n_side_bytes);
/* Run find_all_mems (+ tag queries) for every read of the batch; results stay on the device.
 * (A run whose predecessor on this batch had the same shape is enqueued whole, with buffer sizes taken from that run and the
 * counts kept on the device, and synchronises once at the end: pgx_batch_spec_stats.)
 * `stream` is a hipStream_t; NULL = the batch's own non-blocking stream (every batch has one: its uploads and downloads
 * use it too, so batches driven by different host threads overlap on one device -- upload of one, kernels of another,
 * download of a third).  Asynchronous except for the few scalar read-backs that size intermediate buffers; complete
 * when it returns. */
pgx_status pgx_batch_run(pgx_batch *b, uint64_t min_len, uint64_t min_occ, uint32_t flags, void *stream);
/* Copy the results of the last run to host memory owned by the batch (valid until next run/free) */
pgx_status pgx_batch_result(pgx_batch *b, pgx_result *out);
/* The results of the last run where they are: device pointers into buffers owned by the batch (valid until the next
 * run / upload / free; the run has completed when pgx_batch_run returns).  For consumers that stay on the GPU, e.g.
 * the RCCL exchange of the chromosome-sharded mode, instead of pgx_batch_result's copy to host memory. */
typedef struct {
    uint64_t n_reads, n_mems, n_positions;
    const uint64_t *mem_offsets;    /* n_reads + 1 */
    const pgx_mem *mems;            /* n_mems */
    const uint64_t *tag_run_counts; /* n_mems        (NULL when the run had no PGX_RUN_TAGS) */
    const uint64_t *pos_offsets;    /* n_mems + 1    (NULL ...) */
    const uint64_t *positions;      /* n_positions   (NULL ...) */
} pgx_device_result;
pgx_status pgx_batch_device_result(pgx_batch *b, pgx_device_result *out);
/* Device-side totals of the last run without downloading arrays */
pgx_status pgx_batch_counts(pgx_batch *b, uint64_t *n_mems, uint64_t *n_positions, uint64_t *n_extensions);
pgx_status pgx_batch_timing(pgx_batch *b, pgx_timing *out);
/* Runs of this batch that were sized speculatively (from the totals of the previous run with the same number of reads and the
 * same parameters: no mid-run read-back, one synchronisation at the end), and how many of those had to be repeated with exact
 * sizes because a capacity was too small.  Results never depend on it; PGX_SPEC=0 in the environment switches it off. */
pgx_status pgx_batch_spec_stats(pgx_batch *b, uint32_t *speculative_runs, uint32_t *fallbacks);
void pgx_batch_free(pgx_batch *b);

/* Convenience: create + run + result in one call (what the find_mems CLI uses). */
pgx_status pgx_find_mems_batch(pgx_index *h, int device, const uint8_t *reads, const uint64_t *offsets,
                               uint64_t n_reads, uint64_t min_len, uint64_t min_occ, uint32_t flags,
                               pgx_batch **batch_out, pgx_result *result_out);

/* The same with the reads sharded over several devices of the node (the unit that shards is the per-read loop of
 * src/find_mems.cpp:94-139): n_slices contiguous slices of the batch, slice i on devices[i] (a device may be named more than
 * once) with its own host thread, batch and stream; the index image is replicated, there is no collective.  first_read
 * (n_slices + 1 entries) receives the slice boundaries; results_out[i] / batches_out[i] are those of slice i: concatenated
 * in slice order they are bit-identical to the unsharded result.  Free every batch with pgx_batch_free. */
pgx_status pgx_find_mems_sharded(pgx_index *h, const int *devices, uint32_t n_slices, const uint8_t *reads, const uint64_t *offsets,
                                 uint64_t n_reads, uint64_t min_len, uint64_t min_occ, uint32_t flags, pgx_batch **batches_out,
                                 pgx_result *results_out, uint64_t *first_read);

/* ---- chromosome-sharded mode (BASELINE configs[4]): exchange of per-read MEM lists over RCCL ------------------
 * Every rank (one process per GPU) holds the indexes of some chromosomes ("shards"), searches ALL reads in each of them
 * (one pgx_batch per shard, same reads) and calls pgx_exchange_mems; every rank receives, for every read, the concatenation in
 * shard order of the per-shard MEM lists -- each list bit-exact with the reference run on that shard's index.  (This is not
 * the MEM set of a merged whole-genome index: maximality, BWT coordinates and min_occ are per shard.)  The per-read
 * offsets and the 32-byte records travel device to device: ncclAllGather of u32 offsets, then one ncclBroadcast per rank of
 * exactly that rank's record count (no padding), then a device kernel interleaves the records per read.  RCCL is loaded on
 * first use.  The communicator id comes from rank 0 (pgx_comm_unique_id) and reaches the other ranks by whatever
 * means the caller has (a file, MPI, torch.distributed's store ...). */
#define PGX_COMM_ID_BYTES 128
typedef struct pgx_comm pgx_comm;
pgx_status pgx_comm_unique_id(uint8_t id[PGX_COMM_ID_BYTES]);
pgx_status pgx_comm_init(const uint8_t id[PGX_COMM_ID_BYTES], int rank, int world, int device, pgx_comm **out);
void pgx_comm_free(pgx_comm *c);
typedef struct {
    uint64_t n_reads, n_mems;
    const uint64_t *mem_offsets;  /* n_reads + 1 (device) */
    const pgx_mem *mems;          /* n_mems (device): read by read, within a read shard by shard, within a shard discovery order */
    const uint32_t *shard_of_mem; /* n_mems (device) */
} pgx_exchange_result;
/* batches[k] holds the finished run of shard shard_ids[k] of THIS rank (n_local of them); owner_of_shard[c] (n_shards entries,
 * identical on every rank) names the rank that holds shard c.  Collective: every rank of the communicator calls it.  The
 * result lives in buffers of the communicator until its next exchange. */
pgx_status pgx_exchange_mems(pgx_comm *c, pgx_batch *const *batches, const uint32_t *shard_ids, uint32_t n_local,
                             const uint32_t *owner_of_shard, uint32_t n_shards, pgx_exchange_result *out);
/* The host-side plan of an exchange, as a function of its own (no device, no RCCL: the unit tests of the slot arithmetic call it on the
 * CPU tier).  owner_of_shard as above; gathered = what the metadata all-gather delivers: `world` rows of PGX_XCH_META_HEAD + max_local
 * u64 each -- {status, n_reads, n_shards, digest of owner_of_shard, max_local, MEMs of the rank's 1st, 2nd, ... shard (ascending shard id)}.
 * Outputs (caller-allocated): slot[n_shards] = row of shard c in the gathered offsets (owner * max_local + k for the owner's k-th shard),
 * rec_base[world + 1] = first record of every rank in the gathered record array, src_base[n_shards] = first record of every shard,
 * *n_reads = the number of reads every rank that owns a shard reports.  PGX_ERR_ARG when any rank reported a non-zero status, ranks
 * disagree on n_reads / n_shards / the owner table, or the gathered offsets would exceed PGX_XCH_MAX_OFFSET_BYTES (the caller must
 * then exchange its reads in chunks: at 100 M reads x 24 shards on 8 ranks the offsets alone are 9.6 GB; 16 M reads per call keep
 * them at 1.5 GB). */
#define PGX_XCH_META_HEAD 5u
#define PGX_XCH_MAX_OFFSET_BYTES (4ull << 30)
pgx_status pgx_exchange_plan(uint32_t world, const uint32_t *owner_of_shard, uint32_t n_shards, const uint64_t *gathered,
                             uint32_t *max_local, uint32_t *slot, uint64_t *rec_base, uint64_t *src_base, uint64_t *n_reads);
/* digest of an owner table as the metadata carries it (FNV-1a over the entries) */
uint64_t pgx_exchange_owner_digest(const uint32_t *owner_of_shard, uint32_t n_shards);
/* copy the last exchange's result to host arrays (any may be NULL) */
pgx_status pgx_exchange_download(pgx_comm *c, uint64_t *mem_offsets, pgx_mem *mems, uint32_t *shard_of_mem);

/* find_mems_function (include/pangenome_index/algorithm.hpp:653-736) for n independent (read, start) pairs: query i
 * evaluates the function on read read_of[i] at start position x[i] and reports the start position it returns (next_x),
 * whether it pushed a MEM (has_mem, mem[i]) and the extensions it performed (n_ext, may be NULL).  x[i] > length is
 * undefined in the reference (:658 wraps) and defined here as "return length".  The batch entry points above are the
 * product path; this one serves the compat header's per-call find_mems_function and tests of the state machine. */
pgx_status pgx_find_mems_function_batch(pgx_index *h, int device, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads,
                                        const uint64_t *read_of, const uint64_t *x, uint64_t n, uint64_t min_len, uint64_t min_occ,
                                        uint64_t *next_x, pgx_mem *mem, uint8_t *has_mem, uint64_t *n_ext);

/* device helpers */
pgx_status pgx_device_count(int *n);
pgx_status pgx_device_name(int device, char *buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif

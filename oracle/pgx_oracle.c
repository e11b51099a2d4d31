/*
 * pgx_oracle.c -- CPU ORACLE for the find_mems hot path.  TEST INFRASTRUCTURE ONLY (see header).
 *
 * Literal restatement of the reference's query-side arithmetic, using the reference's own
 * data-structure choices (Elias-Fano predecessor over block starts, 10-run varint blocks,
 * Elias-Fano rank/select for the tag array).  Paths cited below are relative to /root/reference.
 * Parity status: restatement-derived; pinned by fixtures + SURVEY 8c known answers + brute force.
 *
 * Third-party arithmetic that is NOT under /root/reference (versions unpinned there):
 *   - vgteam/sdsl-lite: int_vector / bit_vector / sd_vector serialisation and
 *     sd_vector::{predecessor, rank_1, select_1} semantics (restated from the published layout;
 *     pinned by parsing xy.ri and xy_bidirectional_compressed.tags to the last byte).
 *   - jltsiren/gbwt ByteCode: LSB-first base-128 varint, 0x80 = continue.
 *   - jltsiren/gbwtgraph Position::encode: (id << 11) | (is_rev << 10) | offset.
 */
#include "pgx_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <time.h>

static __thread char g_err[512];
const char *orc_last_error(void) { return g_err; }
#define FAIL(...)                                     \
    do {                                              \
        snprintf(g_err, sizeof g_err, __VA_ARGS__);   \
        goto fail;                                    \
    } while (0)

/* include/pangenome_index/utils.hpp:9-11 */
static const uint8_t NUC[6] = {'\n', 'A', 'C', 'G', 'N', 'T'};

/* ------------------------------------------------------------------------------------------ */
/* byte cursor                                                                                 */
typedef struct {
    const uint8_t *p;
    uint64_t n, o;
    int bad;
} cur_t;

static uint64_t rd_u64(cur_t *c) {
    uint64_t v = 0;
    if (c->o + 8 > c->n) { c->bad = 1; return 0; }
    memcpy(&v, c->p + c->o, 8);
    c->o += 8;
    return v;
}
static uint32_t rd_u32(cur_t *c) {
    uint32_t v = 0;
    if (c->o + 4 > c->n) { c->bad = 1; return 0; }
    memcpy(&v, c->p + c->o, 4);
    c->o += 4;
    return v;
}
static uint8_t rd_u8(cur_t *c) {
    if (c->o + 1 > c->n) { c->bad = 1; return 0; }
    return c->p[c->o++];
}

/* ------------------------------------------------------------------------------------------ */
/* sdsl int_vector<w>: u64 size_in_bits, [u8 width iff w==0], ceil(bits/64) u64 words          */
typedef struct {
    uint64_t bits, n; /* n = number of elements */
    uint8_t width;
    uint64_t *w;
    uint64_t nwords;
} iv_t;

static void iv_free(iv_t *v) { free(v->w); v->w = NULL; }

static int iv_load(cur_t *c, iv_t *v, int fixed_width) {
    memset(v, 0, sizeof *v);
    v->bits = rd_u64(c);
    v->width = fixed_width ? (uint8_t)fixed_width : rd_u8(c);
    if (c->bad) return -1;
    v->nwords = (v->bits + 63) / 64;
    if (c->o + v->nwords * 8 > c->n) { c->bad = 1; return -1; }
    v->w = (uint64_t *)malloc((v->nwords + 1) * 8);
    if (!v->w) return -1;
    memcpy(v->w, c->p + c->o, v->nwords * 8);
    v->w[v->nwords] = 0;
    c->o += v->nwords * 8;
    v->n = v->width ? v->bits / v->width : 0;
    return 0;
}

static inline uint64_t iv_get(const iv_t *v, uint64_t i) {
    uint64_t bit = i * v->width, wd = bit >> 6, sh = bit & 63;
    uint64_t x = v->w[wd] >> sh;
    if (sh + v->width > 64) x |= v->w[wd + 1] << (64 - sh);
    return v->width == 64 ? x : (x & ((1ULL << v->width) - 1));
}
static inline int bv_get(const iv_t *v, uint64_t i) { return (int)((v->w[i >> 6] >> (i & 63)) & 1); }

/* sdsl select_support_mcl: u64 arg_cnt; if nonzero: iv<0> superblock, bit_vector mini_or_long,
 * then one iv<0> (long or mini) per 4096-arg superblock.  Skipped: rebuilt from the bits. */
static int skip_select_support(cur_t *c) {
    uint64_t cnt = rd_u64(c);
    if (c->bad) return -1;
    if (cnt) {
        iv_t t;
        if (iv_load(c, &t, 0)) return -1;
        iv_free(&t);
        if (iv_load(c, &t, 1)) return -1;
        iv_free(&t);
        uint64_t sb = (cnt + 4095) >> 12;
        for (uint64_t i = 0; i < sb; i++) {
            if (iv_load(c, &t, 0)) return -1;
            iv_free(&t);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* sdsl sd_vector<> (Elias-Fano): u64 size, u8 wl, iv<0> low, bit_vector high, 2 select supports */
typedef struct {
    uint64_t size, ones, zeros;
    uint8_t wl;
    iv_t low, high;
    uint64_t *cum1; /* ones before each 512-bit chunk of high */
    uint64_t nchunks;
} ef_t;

static void ef_free(ef_t *e) {
    iv_free(&e->low);
    iv_free(&e->high);
    free(e->cum1);
    e->cum1 = NULL;
}

static int ef_load(cur_t *c, ef_t *e) {
    memset(e, 0, sizeof *e);
    e->size = rd_u64(c);
    e->wl = rd_u8(c);
    if (c->bad) return -1;
    if (iv_load(c, &e->low, 0)) return -1;
    if (iv_load(c, &e->high, 1)) return -1;
    if (skip_select_support(c)) return -1;
    if (skip_select_support(c)) return -1;
    e->nchunks = (e->high.nwords + 7) / 8;
    e->cum1 = (uint64_t *)malloc((e->nchunks + 1) * 8);
    if (!e->cum1) return -1;
    uint64_t acc = 0;
    for (uint64_t k = 0; k < e->nchunks; k++) {
        e->cum1[k] = acc;
        for (uint64_t w = k * 8; w < k * 8 + 8 && w < e->high.nwords; w++)
            acc += (uint64_t)__builtin_popcountll(e->high.w[w]);
    }
    e->cum1[e->nchunks] = acc;
    e->ones = acc;
    e->zeros = e->high.bits - acc;
    return 0;
}

static inline uint64_t word_select(uint64_t w, uint64_t k) { /* position of k-th (0-based) set bit */
    for (uint64_t i = 0; i < k; i++) w &= w - 1;
    return (uint64_t)__builtin_ctzll(w);
}

/* position in high of the k-th (0-based) bit equal to `one` */
static uint64_t ef_high_select(const ef_t *e, uint64_t k, int one) {
    uint64_t lo = 0, hi = e->nchunks; /* last chunk with count_before <= k */
    while (hi - lo > 1) {
        uint64_t mid = (lo + hi) / 2;
        uint64_t before = one ? e->cum1[mid] : mid * 512 - e->cum1[mid];
        if (before <= k) lo = mid; else hi = mid;
    }
    uint64_t before = one ? e->cum1[lo] : lo * 512 - e->cum1[lo];
    for (uint64_t w = lo * 8; w < e->high.nwords; w++) {
        uint64_t x = one ? e->high.w[w] : ~e->high.w[w];
        uint64_t pc = (uint64_t)__builtin_popcountll(x);
        if (before + pc > k) return w * 64 + word_select(x, k - before);
        before += pc;
    }
    return e->high.bits;
}

/* value of the i-th (0-based) one: sdsl sd_vector::select_1(i+1) */
static uint64_t ef_select(const ef_t *e, uint64_t i) {
    uint64_t hp = ef_high_select(e, i, 1) - i;
    return (hp << e->wl) | iv_get(&e->low, i);
}

/* sdsl sd_vector rank_1(i): number of ones in [0, i) */
static uint64_t ef_rank(const ef_t *e, uint64_t i) {
    if (i >= e->size) return e->ones;
    uint64_t hv = i >> e->wl, lv = i & ((1ULL << e->wl) - 1);
    if (hv >= e->zeros) return e->ones;
    uint64_t a = hv == 0 ? 0 : ef_high_select(e, hv - 1, 0) - (hv - 1); /* elements with high < hv */
    uint64_t b = ef_high_select(e, hv, 0) - hv;                         /* elements with high <= hv */
    while (a < b && iv_get(&e->low, a) < lv) a++;
    return a;
}

/* sdsl (vgteam fork) sd_vector::predecessor(i): last one at or before i, as (rank, position).
 * Call site src/r-index.cpp:621.  Positions >= size() resolve to the last one (the reference
 * calls it with pos == bwt_size() for the initial interval). */
static int ef_predecessor(const ef_t *e, uint64_t i, uint64_t *rank, uint64_t *pos) {
    uint64_t r = (i >= e->size) ? e->ones : ef_rank(e, i + 1);
    if (r == 0) return 0;
    *rank = r - 1;
    *pos = ef_select(e, r - 1);
    return 1;
}

/* gbwt::ByteCode::read (call sites src/r-index.cpp:70,148; src/tag_arrays.cpp:819,826) */
static inline uint64_t bytecode_read(const uint8_t *s, uint64_t n, uint64_t *i, int *over) {
    uint64_t off = 0, res = 0;
    for (;;) {
        if (*i >= n) { if (over) *over = 1; return res; }
        uint8_t b = s[(*i)++];
        /* (bits beyond 64 are dropped: only mis-parsed streams -- quirk 3 -- ever get there, a shift by >= 64 is undefined in C) */
        if (b & 0x80) { if (off < 64) res += ((uint64_t)(b & 0x7F)) << off; off += 7; }
        else { if (off < 64) res += ((uint64_t)b) << off; return res; }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* FastLocate                                                                                  */
typedef struct { /* Run_blocks, include/pangenome_index/r-index.hpp:134-297 */
    uint64_t *cum; uint64_t ncum;
    uint64_t *sym, *len; uint64_t nruns;
} lblock_t;

struct orc_ri {
    uint8_t *file; uint64_t file_n, consumed;
    uint32_t tag, version; uint64_t max_length, flags;
    iv_t samples, last_to_run, sym_map_iv, C_iv, enc_starts;
    ef_t last, blocks_start_pos;
    uint8_t sym_map[256];
    uint64_t C[8], sigma, sequence_size;
    /* legacy */
    lblock_t *blocks; uint64_t nblocks;
    /* encoded */
    int encoded, hasN; uint64_t enc_block_size;
    const uint8_t *stream; uint64_t stream_n;
    uint8_t comp[256]; /* initialize_complement_table, src/r-index.cpp:1512-1529 */
};

static uint8_t *read_file(const char *path, uint64_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *b = (uint8_t *)malloc((size_t)sz + 16);
    if (b && fread(b, 1, (size_t)sz, f) != (size_t)sz) { free(b); b = NULL; }
    fclose(f);
    if (b) { memset(b + sz, 0, 16); *n = (uint64_t)sz; }
    return b;
}

void orc_ri_free(orc_ri *r) {
    if (!r) return;
    iv_free(&r->samples); iv_free(&r->last_to_run); iv_free(&r->sym_map_iv); iv_free(&r->C_iv);
    iv_free(&r->enc_starts); ef_free(&r->last); ef_free(&r->blocks_start_pos);
    if (r->blocks) {
        for (uint64_t i = 0; i < r->nblocks; i++) { free(r->blocks[i].cum); free(r->blocks[i].sym); free(r->blocks[i].len); }
        free(r->blocks);
    }
    free(r->file);
    free(r);
}

orc_ri *orc_ri_load(const char *path) {
    orc_ri *r = (orc_ri *)calloc(1, sizeof *r);
    if (!r) return NULL;
    r->file = read_file(path, &r->file_n);
    if (!r->file) FAIL("Cannot open r-index: %s", path);
    cur_t c = {r->file, r->file_n, 0, 0};
    /* Header::load src/r-index.cpp:171-177 ; checks :412-420 */
    r->tag = rd_u32(&c); r->version = rd_u32(&c); r->max_length = rd_u64(&c); r->flags = rd_u64(&c);
    if (c.bad || r->tag != 0x6B3741D8u) FAIL("FastLocate: Invalid tag");
    if (r->version != 1) FAIL("FastLocate: Expected v1, got v%u", r->version);
    /* common prefix: src/r-index.cpp:430-438 (encoded) == :384-393 (legacy) */
    if (iv_load(&c, &r->samples, 0)) FAIL("truncated: samples");
    if (ef_load(&c, &r->last)) FAIL("truncated: last");
    if (iv_load(&c, &r->last_to_run, 0)) FAIL("truncated: last_to_run");
    if (iv_load(&c, &r->sym_map_iv, 8)) FAIL("truncated: sym_map");
    if (iv_load(&c, &r->C_iv, 64)) FAIL("truncated: C");
    if (ef_load(&c, &r->blocks_start_pos)) FAIL("truncated: blocks_start_pos");
    r->sequence_size = rd_u64(&c);
    if (c.bad) FAIL("truncated: sequence_size");
    if (r->sym_map_iv.n != 256) FAIL("sym_map has %llu entries", (unsigned long long)r->sym_map_iv.n);
    for (int i = 0; i < 256; i++) r->sym_map[i] = (uint8_t)iv_get(&r->sym_map_iv, (uint64_t)i);
    r->sigma = r->C_iv.n;
    if (r->sigma < 2 || r->sigma > 6) FAIL("unsupported alphabet size %llu", (unsigned long long)r->sigma);
    for (uint64_t i = 0; i < r->sigma; i++) r->C[i] = iv_get(&r->C_iv, i);
    if (r->flags & 1ULL) { /* ENCODED_BLOCKS, src/r-index.cpp:441-455 */
        r->encoded = 1;
        r->enc_block_size = rd_u64(&c);
        r->hasN = rd_u8(&c) != 0;
        if (iv_load(&c, &r->enc_starts, 0)) FAIL("truncated: blocks_encoded_start_bits");
        r->stream_n = rd_u64(&c);
        if (c.bad || c.o + r->stream_n > c.n) FAIL("truncated: encoded stream");
        r->stream = c.p + c.o;
        c.o += r->stream_n;
        if (r->enc_starts.n == 0) r->encoded = 0; /* is_encoded(), r-index.hpp:409 */
    } else { /* FastLocate::load, src/r-index.cpp:395-402 ; Run_blocks::load hpp:280-290 */
        r->nblocks = rd_u64(&c);
        if (c.bad || r->nblocks > r->file_n) FAIL("truncated: blocks_size");
        r->blocks = (lblock_t *)calloc(r->nblocks ? r->nblocks : 1, sizeof(lblock_t));
        for (uint64_t b = 0; b < r->nblocks; b++) {
            iv_t cum;
            if (iv_load(&c, &cum, 64)) FAIL("truncated: block %llu cum", (unsigned long long)b);
            lblock_t *lb = &r->blocks[b];
            lb->ncum = cum.n;
            lb->cum = (uint64_t *)malloc((cum.n + 1) * 8);
            for (uint64_t i = 0; i < cum.n; i++) lb->cum[i] = iv_get(&cum, i);
            iv_free(&cum);
            lb->nruns = rd_u64(&c);
            if (c.bad || lb->nruns > r->file_n) FAIL("truncated: block %llu runs", (unsigned long long)b);
            lb->sym = (uint64_t *)malloc((lb->nruns + 1) * 8);
            lb->len = (uint64_t *)malloc((lb->nruns + 1) * 8);
            for (uint64_t i = 0; i < lb->nruns; i++) { lb->sym[i] = rd_u64(&c); lb->len[i] = rd_u64(&c); }
            if (c.bad) FAIL("truncated: block %llu", (unsigned long long)b);
        }
    }
    r->consumed = c.o;
    /* initialize_complement_table, src/r-index.cpp:1512-1529 */
    for (int i = 0; i < 256; i++) r->comp[i] = (uint8_t)i;
    r->comp['A'] = 'T'; r->comp['C'] = 'G'; r->comp['G'] = 'C'; r->comp['T'] = 'A';
    r->comp['a'] = 't'; r->comp['c'] = 'g'; r->comp['g'] = 'c'; r->comp['t'] = 'a';
    return r;
fail:
    orc_ri_free(r);
    return NULL;
}

uint64_t orc_ri_bwt_size(const orc_ri *r) { return r->sequence_size; }
uint64_t orc_ri_sigma(const orc_ri *r) { return r->sigma; }
int orc_ri_is_encoded(const orc_ri *r) { return r->encoded; }
int orc_ri_has_N(const orc_ri *r) { return r->hasN; }
uint64_t orc_ri_n_blocks(const orc_ri *r) { return r->encoded ? r->enc_starts.n : r->nblocks; }
uint64_t orc_ri_n_block_starts(const orc_ri *r) { return r->blocks_start_pos.ones; }
uint64_t orc_ri_C(const orc_ri *r, uint64_t i) { return r->C[i]; }
uint8_t orc_ri_sym_map(const orc_ri *r, uint64_t c) { return r->sym_map[c & 255]; }
uint64_t orc_ri_block_start(const orc_ri *r, uint64_t i) { return ef_select(&r->blocks_start_pos, i); }
uint64_t orc_ri_max_length(const orc_ri *r) { return r->max_length; }
uint64_t orc_ri_samples_size(const orc_ri *r) { return r->samples.n; }
uint64_t orc_ri_sample(const orc_ri *r, uint64_t i) { return iv_get(&r->samples, i); }
uint64_t orc_ri_last_ones(const orc_ri *r) { return r->last.ones; }
uint64_t orc_ri_last_size(const orc_ri *r) { return r->last.size; }
uint64_t orc_ri_last_select(const orc_ri *r, uint64_t i) { return ef_select(&r->last, i); }
uint64_t orc_ri_last_to_run(const orc_ri *r, uint64_t i) { return iv_get(&r->last_to_run, i); }
uint64_t orc_ri_encoded_stream_bytes(const orc_ri *r) { return r->stream_n; }
uint64_t orc_ri_file_bytes_consumed(const orc_ri *r) { return r->consumed; }
uint64_t orc_ri_block_nruns(const orc_ri *r, uint64_t b) { return r->blocks ? r->blocks[b].nruns : 0; }
void orc_ri_block_run(const orc_ri *r, uint64_t b, uint64_t i, uint64_t *sym, uint64_t *len) {
    *sym = r->blocks[b].sym[i]; *len = r->blocks[b].len[i];
}
uint64_t orc_ri_block_cum(const orc_ri *r, uint64_t b, uint64_t i) { return r->blocks[b].cum[i]; }

/* Run_blocks::rankAt, include/pangenome_index/r-index.hpp:180-221 */
static uint64_t lblock_rankAt(const lblock_t *b, uint64_t pos, uint64_t symbol) {
    uint64_t rank = 0, run_num = 0, current_position = 0;
    while (run_num < b->nruns) {
        if (b->sym[run_num] == symbol) {
            if (current_position + b->len[run_num] > pos) { rank += (pos - current_position); break; }
            else rank += b->len[run_num];
        }
        current_position += b->len[run_num];
        run_num++;
        if (current_position > pos) break;
    }
    return rank;
}

/* FastLocate::rank_at_cached, src/r-index.cpp:593-603 */
static void rank_at_cached_legacy(const orc_ri *r, uint64_t pos, uint64_t *rv) {
    uint64_t bid = 0, bstart = 0;
    ef_predecessor(&r->blocks_start_pos, pos, &bid, &bstart);
    const lblock_t *b = &r->blocks[bid];
    for (uint64_t i = 0; i < r->sigma; i++)
        rv[i] = lblock_rankAt(b, pos - bstart, NUC[i]) + b->cum[r->sym_map[NUC[i]]];
}

/* EncodedBlock::read_cumulative, src/r-index.cpp:65-82 */
static void enc_read_cumulative(const orc_ri *r, uint64_t *loc, uint64_t cum_len, uint64_t cum_nuc[6]) {
    for (int i = 0; i < 6; i++) cum_nuc[i] = 0;
    for (uint64_t idx = 0; idx < cum_len; idx++) {
        uint64_t val = bytecode_read(r->stream, r->stream_n, loc, NULL);
        for (int i = 0; i < 6; i++)
            if (r->sym_map[NUC[i]] == idx) { cum_nuc[i] = val; break; }
    }
    if (!r->hasN) cum_nuc[4] = 0;
}

/* EncodedBlock::ranks_at, src/r-index.cpp:133-156 */
static void enc_ranks_at(const orc_ri *r, uint64_t loc, uint64_t end_pos, uint64_t rel, uint64_t cum_len, uint64_t out[6]) {
    enc_read_cumulative(r, &loc, cum_len, out);
    uint64_t cur = 0;
    while (loc < end_pos) {
        uint8_t header = r->stream[loc++];
        int code = (header >> 5) & 7;
        uint64_t prefix = header & 0x1F, run_length;
        if (prefix < 31) run_length = prefix + 1;
        else run_length = 32 + bytecode_read(r->stream, r->stream_n, &loc, NULL);
        if (code > 5) code = 5; /* codes 6,7 are never written (symbol_to_code, hpp:664-668) */
        if (cur + run_length > rel) { out[code] += (rel - cur); return; }
        out[code] += run_length;
        cur += run_length;
    }
}

static void enc_counts6(const orc_ri *r, uint64_t pos, uint64_t cum_len, uint64_t counts6[6]) {
    uint64_t bid = 0, bstart = 0;
    ef_predecessor(&r->blocks_start_pos, pos, &bid, &bstart);
    uint64_t loc = iv_get(&r->enc_starts, bid); /* src/r-index.cpp:622 */
    uint64_t end_pos = (bid + 1 < r->enc_starts.n) ? iv_get(&r->enc_starts, bid + 1) : r->stream_n; /* :623 */
    enc_ranks_at(r, loc, end_pos, pos - bstart, cum_len, counts6);
}

/* FastLocate::rank_at_cached_encoded, src/r-index.cpp:619-641 */
void orc_rank_at_cached(const orc_ri *r, uint64_t pos, uint64_t *out) {
    if (!r->encoded) { rank_at_cached_legacy(r, pos, out); return; }
    uint64_t counts6[6];
    enc_counts6(r, pos, r->hasN ? 6 : 5, counts6);
    for (uint64_t i = 0; i < r->sigma; i++) out[i] = counts6[r->sym_map[NUC[i]]];
}

/* True ranks of the six nuc codes (used by ORC_MODE_STRICT and by the device-image tests). */
void orc_rank6_true(const orc_ri *r, uint64_t pos, uint64_t out[6]) {
    if (r->encoded) {
        enc_counts6(r, pos, r->sigma, out); /* the writer emits sigma varints, src/r-index.cpp:339 */
        return;
    }
    uint64_t bid = 0, bstart = 0;
    ef_predecessor(&r->blocks_start_pos, pos, &bid, &bstart);
    const lblock_t *b = &r->blocks[bid];
    for (int i = 0; i < 6; i++) {
        int present = (i == 0) || r->sym_map[NUC[i]] != 0;
        out[i] = present ? lblock_rankAt(b, pos - bstart, NUC[i]) + b->cum[r->sym_map[NUC[i]]] : 0;
    }
}

/* FastLocate::backward_extend_encoded src/r-index.cpp:713-756 (legacy twin :1395-1428) */
static orc_biint bwd_compat(const orc_ri *r, orc_biint in, uint8_t a) {
    uint64_t k = in.forward, k_prime = in.reverse;
    int64_t s = in.size;
    uint64_t b = 0, ks[8] = {0}, kk[8] = {0};
    orc_rank_at_cached(r, k + (uint64_t)s, ks);
    orc_rank_at_cached(r, k, kk);
    const uint8_t comp_sym = r->comp[a];
    const uint64_t comp_idx = r->sym_map[comp_sym];
    while (b < 6 && r->sym_map[NUC[b]] < comp_idx) {
        const uint64_t idx = r->sym_map[r->comp[NUC[b]]];
        k_prime += ks[idx] - kk[idx];
        b++;
    }
    uint64_t rank_ks = ks[r->sym_map[a]], rank_k = kk[r->sym_map[a]];
    orc_biint out = {0, 0, 0};
    if (rank_k >= rank_ks) return out;
    out.size = (int64_t)(rank_ks - rank_k);
    out.forward = rank_k + r->C[r->sym_map[a]];
    out.reverse = k_prime;
    return out;
}

/* Textbook FMD backward extension (Li 2012, alg. 2) over the true ranks; symbol order
 * \n < A < C < G < N < T, complement \n<->\n A<->T C<->G N<->N.  Bytes outside the index
 * alphabet match nothing. */
static orc_biint bwd_strict(const orc_ri *r, orc_biint in, uint8_t a) {
    static const int COMP_CODE[6] = {0, 5, 3, 2, 4, 1};
    orc_biint out = {0, 0, 0};
    int code = -1;
    for (int i = 0; i < 6; i++) if (NUC[i] == a) code = i;
    if (code < 0) return out;
    if (code != 0 && r->sym_map[a] == 0) return out;
    uint64_t ks[6], kk[6];
    orc_rank6_true(r, in.forward + (uint64_t)in.size, ks);
    orc_rank6_true(r, in.forward, kk);
    uint64_t k_prime = in.reverse;
    for (int x = 0; x < COMP_CODE[code]; x++) { /* symbols x < comp(a): add s_{comp(x)} */
        int bb = COMP_CODE[x];
        k_prime += ks[bb] - kk[bb];
    }
    if (kk[code] >= ks[code]) return out;
    out.size = (int64_t)(ks[code] - kk[code]);
    out.forward = kk[code] + r->C[r->sym_map[a]];
    out.reverse = k_prime;
    return out;
}

orc_biint orc_backward_extend(const orc_ri *r, int mode, orc_biint in, uint8_t a) {
    return mode == ORC_MODE_STRICT ? bwd_strict(r, in, a) : bwd_compat(r, in, a);
}

/* FastLocate::forward_extend_encoded src/r-index.cpp:758-764 (legacy :1500-1509) */
orc_biint orc_forward_extend(const orc_ri *r, int mode, orc_biint in, uint8_t a) {
    orc_biint tmp = {in.reverse, in.forward, in.size};
    tmp = orc_backward_extend(r, mode, tmp, r->comp[a]);
    orc_biint out = {tmp.reverse, tmp.forward, tmp.size};
    return out;
}

/* EncodedBlock::rank_of_code, src/r-index.cpp:114-131 */
static uint64_t enc_rank_of_code(const orc_ri *r, uint64_t loc, uint64_t end_pos, int target_code, uint64_t rel) {
    uint64_t rank = 0, cur = 0;
    while (loc < end_pos) {
        uint8_t header = r->stream[loc++];
        int code = (header >> 5) & 7;
        uint64_t prefix = header & 0x1F;
        uint64_t run_length = prefix < 31 ? prefix + 1 : 32 + bytecode_read(r->stream, r->stream_n, &loc, NULL);
        if (code == target_code) {
            if (cur + run_length > rel) { rank += (rel - cur); break; }
            rank += run_length;
        }
        cur += run_length;
        if (cur > rel) break;
    }
    return rank;
}

/* FastLocate::rankAt_encoded, src/r-index.cpp:570-590 (legacy rankAt :558-568) */
static uint64_t rankAt_literal(const orc_ri *r, uint64_t pos, uint8_t symbol) {
    uint64_t bid = 0, bstart = 0;
    ef_predecessor(&r->blocks_start_pos, pos, &bid, &bstart);
    if (!r->encoded) {
        const lblock_t *b = &r->blocks[bid];
        return lblock_rankAt(b, pos - bstart, symbol) + b->cum[r->sym_map[symbol]];
    }
    uint64_t loc = iv_get(&r->enc_starts, bid);
    uint64_t end_pos = (bid + 1 < r->enc_starts.n) ? iv_get(&r->enc_starts, bid + 1) : r->stream_n;
    uint64_t cum[6];
    enc_read_cumulative(r, &loc, 6, cum); /* EncodedBlock blk(..., 6, ...): always six entries (quirk 3) */
    int target_code = 0;                  /* symbol_to_code, r-index.hpp:664-667: unknown -> 0 */
    for (int i = 0; i < 6; i++) if (NUC[i] == symbol) target_code = i;
    return enc_rank_of_code(r, loc, end_pos, target_code, pos - bstart) + cum[target_code];
}

/* One LF step: FastLocate::LF (src/r-index.cpp:650-687) / LF_encoded (:689-711) of the inclusive range
 * [*first, *second] by `sym`; the empty range is {1, 0}.  STRICT: textbook backward-search step over the true ranks. */
void orc_LF(const orc_ri *r, int mode, uint8_t sym, uint64_t *first, uint64_t *second) {
    uint64_t lo = *first, hi = *second;
    if (mode == ORC_MODE_STRICT) {
        int code = -1;
        for (int c = 1; c < 6; c++) if (NUC[c] == sym && r->sym_map[sym] != 0) code = c;
        if (code < 0 || lo > hi) { *first = 1; *second = 0; return; }
        uint64_t a[6], b[6];
        orc_rank6_true(r, lo, a);
        orc_rank6_true(r, hi + 1, b);
        if (b[code] == a[code]) { *first = 1; *second = 0; return; }
        *first = a[code] + r->C[r->sym_map[sym]];
        *second = *first + (b[code] - a[code]) - 1;
        return;
    }
    if (!r->encoded && !r->sym_map[sym]) { *first = 1; *second = 0; return; } /* :653 */
    if (lo > hi) { *first = 1; *second = 0; return; }                          /* :655-657, :691 */
    uint64_t f = rankAt_literal(r, lo, sym);
    uint64_t inside = rankAt_literal(r, hi + 1, sym) - f;
    if (inside == 0) { *first = 1; *second = 0; return; }
    *first = f + r->C[r->sym_map[sym]];
    *second = *first + inside - 1;
}

/* FastLocate::count / count_encoded, include/pangenome_index/r-index.hpp:540-556: LF per symbol from the end */
void orc_count(const orc_ri *r, int mode, const uint8_t *read, uint64_t len, uint64_t *first, uint64_t *second) {
    uint64_t lo = 0, hi = r->sequence_size - 1; /* {0, bwt_size() - 1}, r-index.hpp:541,551 */
    for (uint64_t i = len; i > 0; i--) orc_LF(r, mode, read[i - 1], &lo, &hi);
    *first = lo;
    *second = hi;
}

/* ------------------------------------------------------------------------------------------ */
/* locate: locateFirst / locateNext / locate / locate_encoded / decompressSA / decompressDA */

/* FastLocate::locateFirst = getSample(0), include/pangenome_index/r-index.hpp:490-492,616-618 */
uint64_t orc_locate_first(const orc_ri *r) { return r->samples.n ? iv_get(&r->samples, 0) : ORC_NO_POSITION; }

/* FastLocate::locateNext, src/r-index.cpp:1363-1366:
 *   iter = last.predecessor(prev); return samples[last_to_run[iter->first] + 1] + (prev - iter->second)
 * Undefined in the reference when no tail sample precedes prev or when the tail belongs to the last run
 * (samples[r] is read past the end); defined here as ORC_NO_POSITION. */
uint64_t orc_locate_next(const orc_ri *r, uint64_t prev) {
    uint64_t rank = 0, pos = 0;
    if (prev == ORC_NO_POSITION || !ef_predecessor(&r->last, prev, &rank, &pos)) return ORC_NO_POSITION;
    const uint64_t run = iv_get(&r->last_to_run, rank) + 1;
    if (run >= r->samples.n) return ORC_NO_POSITION;
    return iv_get(&r->samples, run) + (prev - pos);
}

uint64_t orc_seq_id(const orc_ri *r, uint64_t v) { return v / r->max_length; }      /* r-index.hpp:429 */
uint64_t orc_seq_offset(const orc_ri *r, uint64_t v) { return v % r->max_length; }  /* r-index.hpp:431 */

/* FastLocate::decompressSA, src/r-index.cpp:1343-1353 (decompressDA :1355-1361 divides by max_length) */
void orc_decompress_sa(const orc_ri *r, uint64_t *out) {
    if (!r->sequence_size) return;
    out[0] = orc_locate_first(r);
    for (uint64_t i = 1; i < r->sequence_size; i++) out[i] = orc_locate_next(r, out[i - 1]);
}

/* run holding BWT position pos and the BWT offset of its first symbol.
 * legacy: Run_blocks::run_id_at (r-index.hpp:224-234) as called at src/r-index.cpp:1267-1270;
 * encoded: the inline scan of locate_encoded (:1308-1326) == run_id_and_offset_at (:1189-1213), whose
 * EncodedBlock::skip_header (:83-88) always skips SIX varints (quirk 3).  STRICT skips the sigma varints the
 * writer emitted (:339).  Returns 0 when the literal scan leaves the encoded stream (undefined in the reference). */
static int run_id_and_offset(const orc_ri *r, int mode, uint64_t pos, uint64_t *run_id, uint64_t *off) {
    uint64_t bid = 0, bstart = 0;
    if (!ef_predecessor(&r->blocks_start_pos, pos, &bid, &bstart)) return 0;
    if (!r->encoded) {
        const lblock_t *b = &r->blocks[bid];
        uint64_t cur = 0, run_num = 0;
        while (run_num < b->nruns) {
            if (cur + b->len[run_num] > pos - bstart) break;
            cur += b->len[run_num];
            run_num++;
        }
        *run_id = bid * 10 + run_num; /* block_size = 10, r-index.hpp:312 */
        *off = bstart + cur;
        return 1;
    }
    uint64_t loc = iv_get(&r->enc_starts, bid);
    const uint64_t skip = (mode == ORC_MODE_STRICT) ? r->sigma : 6;
    for (uint64_t i = 0; i < skip; i++) {
        int over = 0;
        (void)bytecode_read(r->stream, r->stream_n, &loc, &over);
        if (over) return 0;
    }
    uint64_t cur = 0, runnum = 0;
    for (;;) {
        if (loc >= r->stream_n) return 0;
        uint8_t header = r->stream[loc++];
        uint64_t prefix = header & 0x1F, run_length;
        if (prefix < 31) run_length = prefix + 1;
        else {
            int over = 0;
            run_length = 32 + bytecode_read(r->stream, r->stream_n, &loc, &over);
            if (over) return 0;
        }
        if (cur + run_length > pos - bstart) {
            *run_id = bid * (r->enc_block_size ? r->enc_block_size : 10) + runnum;
            *off = bstart + cur;
            return 1;
        }
        cur += run_length;
        runnum++;
    }
}

/* the suffix-array values of BWT positions [first, last] in BWT order, as FastLocate::locate (src/r-index.cpp:1252-1290)
 * / locate_encoded (:1299-1341) produce them before seqId / sort / unique.  Returns the number of values
 * (0 for an empty state), or ORC_NO_POSITION when the literal run scan is undefined (see run_id_and_offset). */
uint64_t orc_locate_sa(const orc_ri *r, int mode, uint64_t first, uint64_t last, uint64_t *out) {
    if (last < first) return 0; /* :1255 */
    uint64_t run_id = 0, off = 0;
    if (!run_id_and_offset(r, mode, first, &run_id, &off)) return ORC_NO_POSITION;
    if (run_id >= r->samples.n) return ORC_NO_POSITION;
    uint64_t v = iv_get(&r->samples, run_id); /* getSample(run_id) */
    while (off < first) { v = orc_locate_next(r, v); off++; }
    uint64_t k = 0;
    out[k++] = v;
    for (uint64_t i = first + 1; i <= last; i++) { v = orc_locate_next(r, v); out[k++] = v; }
    return k;
}

typedef struct { orc_mem *v; uint64_t n, cap, total; } memsink_t;
static inline void sink_push(memsink_t *s, orc_mem m) {
    if (s->n < s->cap) s->v[s->n++] = m;
    s->total++;
}

/* find_mems_function, include/pangenome_index/algorithm.hpp:653-736.  pattern[len] reads as '\0'
 * (std::string::operator[] at size()), SURVEY 8a quirk 4. */
static uint64_t find_mems_function(const orc_ri *r, int mode, const uint8_t *pattern, uint64_t len,
                                   uint64_t min_len, uint64_t min_occ, uint64_t x, memsink_t *out,
                                   uint64_t *n_ext) {
#define PAT(j) ((j) < len ? pattern[(j)] : (uint8_t)0)
#define SMALL(b) ((uint64_t)(b).size < min_occ || (b).size <= 0)
    uint64_t j;
    if (len - x < min_len) return len;
    orc_biint bint = {0, 0, (int64_t)r->sequence_size};
    for (j = x + min_len - 1; (int64_t)j >= (int64_t)x; --j) { /* :666 */
        bint = orc_backward_extend(r, mode, bint, PAT(j));
        (*n_ext)++;
        if (SMALL(bint)) return j + 1;
        if (j == 0) break;
    }
    orc_biint bint2 = bint;
    for (j = x + min_len; j < len; ++j) { /* :685 */
        bint = orc_forward_extend(r, mode, bint, PAT(j));
        (*n_ext)++;
        if (SMALL(bint)) break;
        bint2 = bint;
    }
    uint64_t e = j;
    orc_mem m = {x, e, bint2.forward, bint2.size}; /* :713 */
    sink_push(out, m);
    orc_biint back = {0, 0, (int64_t)r->sequence_size};
    for (j = e; j > x; --j) { /* :722 */
        back = orc_backward_extend(r, mode, back, PAT(j));
        (*n_ext)++;
        if (SMALL(back)) return j + 1;
    }
    return j + 1;
#undef PAT
#undef SMALL
}

/* one call of find_mems_function (algorithm.hpp:653-736) at start x <= len: returns the next start; *has_mem / *mem = the MEM it pushed */
uint64_t orc_find_mems_function(const orc_ri *r, int mode, const uint8_t *read, uint64_t len, uint64_t min_len, uint64_t min_occ,
                                uint64_t x, orc_mem *mem, int *has_mem, uint64_t *n_ext) {
    orc_mem tmp;
    memsink_t s = {&tmp, 0, 1, 0};
    uint64_t dummy = 0;
    if (!n_ext) n_ext = &dummy;
    const uint64_t nx = find_mems_function(r, mode, read, len, min_len, min_occ, x, &s, n_ext);
    if (has_mem) *has_mem = s.total ? 1 : 0;
    if (mem && s.total) *mem = tmp;
    return nx;
}

/* find_all_mems, include/pangenome_index/algorithm.hpp:739-757 */
uint64_t orc_find_all_mems(const orc_ri *r, int mode, const uint8_t *read, uint64_t len,
                           uint64_t min_len, uint64_t min_occ, orc_mem *out, uint64_t cap,
                           uint64_t *n_ext) {
    memsink_t s = {out, 0, cap, 0};
    uint64_t dummy = 0, x = 0;
    if (!n_ext) n_ext = &dummy;
    while (x < len) x = find_mems_function(r, mode, read, len, min_len, min_occ, x, &s, n_ext);
    return s.total;
}

/* ------------------------------------------------------------------------------------------ */
/* TagArray                                                                                    */
struct orc_tags {
    uint8_t *file; uint64_t file_n, consumed;
    int format;
    const uint8_t *stream; uint64_t stream_n; /* ByteCode runs */
    iv_t items;                               /* compact runs  */
    ef_t starts, bwt_intervals;
    uint64_t n_items;
};

void orc_tags_free(orc_tags *t) {
    if (!t) return;
    iv_free(&t->items); ef_free(&t->starts); ef_free(&t->bwt_intervals);
    free(t->file);
    free(t);
}

orc_tags *orc_tags_load(const char *path, int format) {
    orc_tags *t = (orc_tags *)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->format = format;
    t->file = read_file(path, &t->file_n);
    if (!t->file) FAIL("Cannot open tag array: %s", path);
    cur_t c = {t->file, t->file_n, 0, 0};
    if (format == ORC_TAGS_BYTECODE) { /* load_compressed_tags, src/tag_arrays.cpp:739-763 */
        t->stream_n = rd_u64(&c);
        if (c.bad || c.o + t->stream_n > c.n) FAIL("truncated: encoded_runs");
        t->stream = c.p + c.o;
        c.o += t->stream_n;
        uint64_t i = 0;
        while (i < t->stream_n) { bytecode_read(t->stream, t->stream_n, &i, NULL); t->n_items++; }
    } else if (format == ORC_TAGS_COMPACT) { /* load_compressed_tags_sdsl, src/tag_arrays.cpp:766-776 */
        if (iv_load(&c, &t->items, 0)) FAIL("truncated: encoded_runs_iv");
        t->n_items = t->items.n;
    } else FAIL("unknown tag format %d", format);
    if (ef_load(&c, &t->starts)) FAIL("truncated: encoded_runs_starts_sd");
    if (ef_load(&c, &t->bwt_intervals)) FAIL("truncated: bwt_intervals");
    t->consumed = c.o;
    return t;
fail:
    orc_tags_free(t);
    return NULL;
}

uint64_t orc_tags_n_runs(const orc_tags *t) { return t->bwt_intervals.ones; }
uint64_t orc_tags_bwt_intervals_size(const orc_tags *t) { return t->bwt_intervals.size; }
uint64_t orc_tags_n_items(const orc_tags *t) { return t->n_items; }
uint64_t orc_tags_n_starts(const orc_tags *t) { return t->starts.ones; }
uint64_t orc_tags_start(const orc_tags *t, uint64_t i) { return ef_select(&t->starts, i); }
uint64_t orc_tags_interval(const orc_tags *t, uint64_t i) { return ef_select(&t->bwt_intervals, i); }
uint64_t orc_tags_file_bytes_consumed(const orc_tags *t) { return t->consumed; }
uint64_t orc_tags_item(const orc_tags *t, uint64_t i) {
    if (t->format == ORC_TAGS_COMPACT) return iv_get(&t->items, i);
    uint64_t loc = 0, v = 0;
    for (uint64_t k = 0; k <= i; k++) v = bytecode_read(t->stream, t->stream_n, &loc, NULL);
    return v;
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

/* TagArray::query_compressed (src/tag_arrays.cpp:780-854) and query_compressed_compact (:856-890),
 * minus the printing.  decode_run (:59-70, length_bits = 9) / decode_run_length_compact (:47-55)
 * followed by gbwtgraph::Position::encode. */
static uint64_t tags_query_into(const orc_tags *t, uint64_t start, uint64_t end, uint64_t *run_nums_out,
                                uint64_t **buf, uint64_t *bufcap, int *overflow) {
    uint64_t first_bit_index = ef_rank(&t->bwt_intervals, start + 1);
    uint64_t end_bit_index = ef_rank(&t->bwt_intervals, end + 1);
    uint64_t run_nums = end_bit_index - first_bit_index + 1;
    *run_nums_out = run_nums;
    if (run_nums > *bufcap) {
        *bufcap = run_nums * 2;
        *buf = (uint64_t *)realloc(*buf, *bufcap * 8);
    }
    uint64_t *up = *buf, n = 0;
    uint64_t current_tag_run_index = first_bit_index - (first_bit_index % 10);
    uint64_t move_tags = first_bit_index % 10;
    uint64_t sel = current_tag_run_index / 10; /* select_1(sel + 1) */
    int over = 0;
    uint64_t loc;
    if (sel >= t->starts.ones) { over = 1; loc = (t->format == ORC_TAGS_COMPACT) ? t->n_items : t->stream_n; }
    else loc = ef_select(&t->starts, sel);
    if (t->format == ORC_TAGS_BYTECODE) {
        while (move_tags > 1) { (void)bytecode_read(t->stream, t->stream_n, &loc, &over); move_tags--; }
        while (run_nums > 0) {
            uint64_t decc = bytecode_read(t->stream, t->stream_n, &loc, &over);
            uint64_t off = decc & 0x3FF, rev = (decc >> 10) & 1, node = decc >> (11 + 9);
            run_nums--;
            up[n++] = (node << 11) | (rev << 10) | off;
        }
    } else {
        while (move_tags > 1) { loc++; move_tags--; }
        while (run_nums > 0) {
            uint64_t enc = 0;
            if (loc < t->n_items) enc = iv_get(&t->items, loc); else over = 1;
            loc++;
            uint64_t off = enc & 0x3FF, rev = (enc >> 10) & 1, node = enc >> 11;
            run_nums--;
            up[n++] = (node << 11) | (rev << 10) | off;
        }
    }
    qsort(up, n, 8, cmp_u64);
    uint64_t u = 0;
    for (uint64_t i = 0; i < n; i++) if (i == 0 || up[i] != up[i - 1]) up[u++] = up[i];
    if (overflow) *overflow = over;
    return u;
}

uint64_t orc_tags_query(const orc_tags *t, uint64_t start, uint64_t end, uint64_t *run_nums,
                        uint64_t *out, uint64_t cap, int *overflow) {
    uint64_t *buf = NULL, bufcap = 0, rn = 0;
    uint64_t u = tags_query_into(t, start, end, &rn, &buf, &bufcap, overflow);
    if (run_nums) *run_nums = rn;
    for (uint64_t i = 0; i < u && i < cap; i++) out[i] = buf[i];
    free(buf);
    return u;
}

/* ------------------------------------------------------------------------------------------ */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_batch_free(orc_batch_result *res) {
    if (!res) return;
    free(res->mem_offsets); free(res->mems); free(res->tag_run_counts); free(res->pos_offsets); free(res->positions);
    free(res);
}

/* The per-read loop of src/find_mems.cpp:94-139 with the printing removed; reads are independent
 * (algorithm.hpp:739 carries no cross-read state) so threads>1 parallelises over reads. */
orc_batch_result *orc_find_mems_batch(const orc_ri *r, const orc_tags *t, int mode,
                                      const uint8_t *reads, const uint64_t *offsets,
                                      uint64_t n_reads, uint64_t min_len, uint64_t min_occ,
                                      int threads) {
    orc_batch_result *res = (orc_batch_result *)calloc(1, sizeof *res);
    res->n_reads = n_reads;
    res->mem_offsets = (uint64_t *)calloc(n_reads + 1, 8);
    if (threads < 1) threads = 1;
    /* pass 1: MEMs.  Per-read result lists are kept in per-thread arenas, then stitched in order. */
    orc_mem **per_read = (orc_mem **)calloc(n_reads ? n_reads : 1, sizeof(orc_mem *));
    uint64_t n_ext_total = 0;
    double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads) reduction(+ : n_ext_total)
    for (int64_t i = 0; i < (int64_t)n_reads; i++) {
        const uint8_t *p = reads + offsets[i];
        uint64_t len = offsets[i + 1] - offsets[i];
        orc_mem small[16];
        uint64_t ne = 0;
        uint64_t cnt = orc_find_all_mems(r, mode, p, len, min_len, min_occ, small, 16, &ne);
        orc_mem *v = NULL;
        if (cnt) {
            v = (orc_mem *)malloc(cnt * sizeof(orc_mem));
            if (cnt <= 16) memcpy(v, small, cnt * sizeof(orc_mem));
            else { uint64_t ne2 = 0; orc_find_all_mems(r, mode, p, len, min_len, min_occ, v, cnt, &ne2); }
        }
        per_read[i] = v;
        res->mem_offsets[i + 1] = cnt;
        n_ext_total += ne;
    }
    res->seconds_mems = now_s() - t0;
    res->n_extensions = n_ext_total;
    for (uint64_t i = 0; i < n_reads; i++) res->mem_offsets[i + 1] += res->mem_offsets[i];
    uint64_t n_mems = res->mem_offsets[n_reads];
    res->mems = (orc_mem *)malloc((n_mems ? n_mems : 1) * sizeof(orc_mem));
    for (uint64_t i = 0; i < n_reads; i++) {
        uint64_t c = res->mem_offsets[i + 1] - res->mem_offsets[i];
        if (c) memcpy(res->mems + res->mem_offsets[i], per_read[i], c * sizeof(orc_mem));
        free(per_read[i]);
    }
    free(per_read);
    if (!t) return res;
    /* pass 2: tag queries, src/find_mems.cpp:129 */
    res->tag_run_counts = (uint64_t *)calloc(n_mems ? n_mems : 1, 8);
    res->pos_offsets = (uint64_t *)calloc(n_mems + 1, 8);
    uint64_t **per_mem = (uint64_t **)calloc(n_mems ? n_mems : 1, sizeof(uint64_t *));
    uint64_t n_over = 0;
    t0 = now_s();
#pragma omp parallel num_threads(threads) reduction(+ : n_over)
    {
        uint64_t *buf = NULL, bufcap = 0;
#pragma omp for schedule(dynamic, 256)
        for (int64_t m = 0; m < (int64_t)n_mems; m++) {
            const orc_mem *mm = &res->mems[m];
            int over = 0;
            uint64_t rn = 0;
            uint64_t u = tags_query_into(t, mm->bwt_start, mm->bwt_start + (uint64_t)mm->size - 1, &rn, &buf, &bufcap, &over);
            res->tag_run_counts[m] = rn;
            res->pos_offsets[m + 1] = u;
            per_mem[m] = (uint64_t *)malloc((u ? u : 1) * 8);
            memcpy(per_mem[m], buf, u * 8);
            n_over += (uint64_t)over;
        }
        free(buf);
    }
    res->seconds_tags = now_s() - t0;
    res->n_tag_overflow = n_over;
    for (uint64_t m = 0; m < n_mems; m++) res->pos_offsets[m + 1] += res->pos_offsets[m];
    res->positions = (uint64_t *)malloc((res->pos_offsets[n_mems] ? res->pos_offsets[n_mems] : 1) * 8);
    for (uint64_t m = 0; m < n_mems; m++) {
        memcpy(res->positions + res->pos_offsets[m], per_mem[m], (res->pos_offsets[m + 1] - res->pos_offsets[m]) * 8);
        free(per_mem[m]);
    }
    free(per_mem);
    return res;
}

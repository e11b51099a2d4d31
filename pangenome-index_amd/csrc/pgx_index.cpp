// pgx_index.cpp -- file parsing and device-image construction (host, run once per index).
//
// Replaces the loaders FastLocate::load_encoded / load (src/r-index.cpp:378-459) and
// TagArray::load_compressed_tags{,_sdsl} (src/tag_arrays.cpp:739-776): instead of rebuilding SDSL
// objects it normalises both .ri layouts to a list of logical BWT runs and packs them into the
// flat image of pgx_image.h.  No query arithmetic lives here.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include "pgx_host.hpp"

namespace pgx {

static thread_local std::string g_last_error;
void set_last_error(const std::string &m) { g_last_error = m; }
const std::string &last_error() { return g_last_error; }

static inline unsigned bytecode_len(uint64_t v) {
    unsigned k = 1;
    while (v > 0x7F) { v >>= 7; k++; }
    return k;
}

// ------------------------------------------------------------------------------------------
void RiFile::parse(const uint8_t *p, uint64_t n) {
    ByteReader r(p, n);
    // Header::load src/r-index.cpp:171-177, checks :412-420
    tag = r.get<uint32_t>("header");
    version = r.get<uint32_t>("header");
    max_length = r.get<uint64_t>("header");
    flags = r.get<uint64_t>("header");
    if (tag != 0x6B3741D8u) throw Error(PGX_ERR_FORMAT, "FastLocate: Invalid tag");
    if (version != 1) throw Error(PGX_ERR_FORMAT, "FastLocate: Expected v1, got v" + std::to_string(version));
    samples.read(r, 0, "samples");
    last.read(r, "last");
    last_to_run.read(r, 0, "last_to_run");
    IntVector sm, Cv;
    sm.read(r, 8, "sym_map");
    Cv.read(r, 64, "C");
    blocks_start_pos.read(r, "blocks_start_pos");
    sequence_size = r.get<uint64_t>("sequence_size");
    if (sm.size() != 256) throw Error(PGX_ERR_FORMAT, "FastLocate: sym_map must have 256 entries");
    for (int i = 0; i < 256; i++) sym_map[i] = (uint8_t)sm.get((uint64_t)i);
    C.resize(Cv.size());
    for (uint64_t i = 0; i < C.size(); i++) C[i] = Cv.get(i);
    const uint64_t sigma = C.size();
    if (sigma < 2 || sigma > 6) throw Error(PGX_ERR_UNSUPPORTED, "alphabet size " + std::to_string(sigma) + " not in 2..6");
    if (sequence_size >> PGX_COUNT_BITS) throw Error(PGX_ERR_UNSUPPORTED, "BWT longer than 2^40 symbols");
    // every indexed symbol must be one of nuc (symbol_to_code, r-index.hpp:664-668)
    uint64_t present = 0;
    for (int c = 0; c < 256; c++) {
        bool in_index = sym_map[c] != 0 || c == '\n';
        if (sym_map[c] >= sigma) throw Error(PGX_ERR_FORMAT, "sym_map value out of range");
        if (sym_map[c] != 0 && code_of_byte((uint8_t)c) < 0)
            throw Error(PGX_ERR_UNSUPPORTED, "indexed symbol outside {\\n,A,C,G,N,T}");
        if (in_index && code_of_byte((uint8_t)c) >= 0) present++;
    }
    if (present != sigma) throw Error(PGX_ERR_UNSUPPORTED, "sym_map and C disagree on the alphabet");
    if (sym_map[(uint8_t)'\n'] != 0) throw Error(PGX_ERR_UNSUPPORTED, "endmarker must map to 0");

    blocks.clear();
    double enc_bytes = 0;
    if (flags & 1ULL) { // ENCODED_BLOCKS, src/r-index.cpp:441-455
        encoded = true;
        enc_block_size = r.get<uint64_t>("encoded_block_size");
        hasN = r.get<uint8_t>("encoded_has_N") != 0;
        IntVector starts;
        starts.read(r, 0, "blocks_encoded_start_bits");
        uint64_t nbytes = r.get<uint64_t>("blocks_encoded_stream_size");
        r.need(nbytes, "blocks_encoded_stream");
        const uint8_t *s = r.p + r.o;
        r.o += nbytes;
        if (starts.size() == 0) throw Error(PGX_ERR_UNSUPPORTED, "encoded index without blocks");
        // rank_at_cached_encoded reads hasN?6:5 cumulative varints (src/r-index.cpp:624) while the
        // writer emits sigma of them (:339): any other combination mis-parses in the reference.
        if ((hasN ? 6u : 5u) != sigma)
            throw Error(PGX_ERR_UNSUPPORTED, "encoded index with sigma != (hasN ? 6 : 5): the reference mis-parses it");
        n_file_blocks = starts.size();
        for (uint64_t b = 0; b < n_file_blocks; b++) {
            uint64_t loc = starts.get(b), end = (b + 1 < n_file_blocks) ? starts.get(b + 1) : nbytes;
            if (loc > end || end > nbytes) throw Error(PGX_ERR_FORMAT, "encoded block offsets not monotone");
            RefBlock blk;
            blk.cum.resize(sigma);
            // a trailing never-filled block (total_runs % 10 == 0) holds 8 zero varints and no runs
            // (Run_blocks() default, r-index.hpp:144); no query ever reaches it (quirk 9)
            if (b >= blocks_start_pos.ones.size()) { blocks.push_back(std::move(blk)); continue; }
            for (uint64_t i = 0; i < sigma && loc < end; i++) blk.cum[i] = bytecode_read(s, end, loc, "encoded block header");
            while (loc < end) {
                uint8_t h = s[loc++];
                uint8_t code = (h >> 5) & 7;
                uint64_t prefix = h & 0x1F;
                uint64_t len = prefix < 31 ? prefix + 1 : 32 + bytecode_read(s, end, loc, "encoded run");
                if (code > 5) throw Error(PGX_ERR_FORMAT, "encoded run with code > 5");
                blk.runs.emplace_back(code, len);
            }
            blocks.push_back(std::move(blk));
        }
        if (!hasN) { // the scan rankAt_encoded performs on this shape (quirk 3): six varints skipped, then headers wherever that lands
            auto vread = [&](uint64_t &loc) { // gbwt::ByteCode::read without a bound but the stream's
                uint64_t off = 0, res = 0;
                for (;;) {
                    if (loc >= nbytes) return res;
                    const uint8_t b = s[loc++];
                    if (off < 64) res += (uint64_t)(b & 0x7F) << off;
                    if (!(b & 0x80)) return res;
                    off += 7;
                }
            };
            lit_runs.assign(blocks_start_pos.ones.size(), {});
            for (uint64_t b = 0; b < lit_runs.size() && b < n_file_blocks; b++) {
                uint64_t loc = starts.get(b);
                const uint64_t end = (b + 1 < n_file_blocks) ? starts.get(b + 1) : nbytes;
                for (int i = 0; i < 6; i++) (void)vread(loc);
                while (loc < end) {
                    const uint8_t hd = s[loc++];
                    const uint64_t code = (hd >> 5) & 7, prefix = hd & 0x1F;
                    const uint64_t len = prefix < 31 ? prefix + 1 : 32 + vread(loc);
                    lit_runs[b].push_back((code << 56) | (len & ((1ull << 56) - 1)));
                }
            }
        }
        enc_bytes = (double)nbytes;
    } else { // FastLocate::load src/r-index.cpp:395-402 ; Run_blocks::load r-index.hpp:280-290
        encoded = false;
        n_file_blocks = r.get<uint64_t>("blocks_size");
        if (n_file_blocks > n) throw Error(PGX_ERR_FORMAT, "blocks_size larger than the file");
        for (uint64_t b = 0; b < n_file_blocks; b++) {
            IntVector cum;
            cum.read(r, 64, "block cumulative ranks");
            RefBlock blk;
            blk.cum.resize(cum.size());
            for (uint64_t i = 0; i < cum.size(); i++) blk.cum[i] = cum.get(i);
            uint64_t nruns = r.get<uint64_t>("runs_size");
            r.need(nruns * 16, "block runs");
            for (uint64_t i = 0; i < nruns; i++) {
                uint64_t sym = r.get<uint64_t>("run"), len = r.get<uint64_t>("run");
                int code = sym < 256 ? code_of_byte((uint8_t)sym) : -1;
                if (code < 0) throw Error(PGX_ERR_UNSUPPORTED, "BWT symbol outside {\\n,A,C,G,N,T}");
                blk.runs.emplace_back((uint8_t)code, len);
            }
            for (uint64_t i = 0; i < sigma; i++) enc_bytes += bytecode_len(i < blk.cum.size() ? blk.cum[i] : 0);
            for (auto &ru : blk.runs) enc_bytes += 1 + (ru.second >= 32 ? bytecode_len(ru.second - 32) : 0);
            blocks.push_back(std::move(blk));
        }
    }
    if (r.o != n) throw Error(PGX_ERR_FORMAT, "trailing bytes after the r-index");
    // drop never-filled trailing blocks and validate against blocks_start_pos
    while (!blocks.empty() && blocks.back().runs.empty()) blocks.pop_back();
    if (blocks.size() != blocks_start_pos.ones.size())
        throw Error(PGX_ERR_FORMAT, "blocks_start_pos does not match the number of non-empty blocks");
    std::vector<uint64_t> acc(sigma, 0);
    uint64_t pos = 0;
    for (uint64_t b = 0; b < blocks.size(); b++) {
        if (blocks_start_pos.ones[b] != pos) throw Error(PGX_ERR_FORMAT, "block start position mismatch");
        for (uint64_t i = 0; i < sigma; i++)
            if (blocks[b].cum.size() < sigma || blocks[b].cum[i] != acc[i])
                throw Error(PGX_ERR_FORMAT, "block cumulative ranks do not match the runs");
        for (auto &ru : blocks[b].runs) {
            if (ru.second == 0) throw Error(PGX_ERR_FORMAT, "zero-length run");
            uint8_t sm_idx = sym_map[kNuc[ru.first]];
            if (ru.first != 0 && sm_idx == 0) throw Error(PGX_ERR_FORMAT, "run symbol not in the alphabet");
            acc[sm_idx] += ru.second;
            pos += ru.second;
        }
    }
    if (pos != sequence_size) throw Error(PGX_ERR_FORMAT, "runs do not add up to sequence_size");
    ref_block_mean_bytes = blocks.empty() ? 0.0 : enc_bytes / (double)blocks.size();
}

// ------------------------------------------------------------------------------------------
static bool try_parse_tags(TagFile &tf, const uint8_t *p, uint64_t n, uint32_t fmt) {
    try {
        ByteReader r(p, n);
        tf.items.clear();
        if (fmt == PGX_TAGS_BYTECODE) { // load_compressed_tags, src/tag_arrays.cpp:739-763
            uint64_t nbytes = r.get<uint64_t>("encoded_runs size");
            r.need(nbytes, "encoded_runs");
            const uint8_t *s = r.p + r.o;
            uint64_t i = 0;
            std::vector<uint64_t> byte_off;
            while (i < nbytes) {
                byte_off.push_back(i);
                tf.items.push_back(bytecode_read(s, nbytes, i, "tag run"));
            }
            r.o += nbytes;
            tf.starts.read(r, "encoded_runs_starts_sd");
            tf.bwt_intervals.read(r, "bwt_intervals");
            if (r.o != n) return false;
            // starts hold the byte offset of every 10th run (encoded_start_every_k_run, tag_arrays.hpp:120)
            for (uint64_t k = 0; k < tf.starts.ones.size(); k++) {
                if (k * 10 >= byte_off.size() || byte_off[k * 10] != tf.starts.ones[k])
                    throw Error(PGX_ERR_UNSUPPORTED, "tag run sampling is not 'every 10th run'");
                tf.starts.ones[k] = k * 10; // normalise to item indexes
            }
        } else { // load_compressed_tags_sdsl, src/tag_arrays.cpp:766-776
            IntVector iv;
            iv.read(r, 0, "encoded_runs_iv");
            if (iv.width == 0) return false;
            tf.starts.read(r, "encoded_runs_starts_sd");
            tf.bwt_intervals.read(r, "bwt_intervals");
            if (r.o != n) return false;
            tf.items.resize(iv.size());
            for (uint64_t i = 0; i < tf.items.size(); i++) tf.items[i] = iv.get(i);
            for (uint64_t k = 0; k < tf.starts.ones.size(); k++)
                if (tf.starts.ones[k] != k * 10) throw Error(PGX_ERR_UNSUPPORTED, "tag run sampling is not 'every 10th run'");
        }
        if (tf.starts.ones.size() != (tf.items.size() + 9) / 10)
            throw Error(PGX_ERR_UNSUPPORTED, "tag run sampling is not 'every 10th run'");
        tf.format = fmt;
        return true;
    } catch (const Error &e) {
        if (e.code == PGX_ERR_UNSUPPORTED) throw;
        return false;
    }
}

void TagFile::parse(const uint8_t *p, uint64_t n, uint32_t format_hint) {
    if (format_hint == PGX_TAGS_BYTECODE || format_hint == PGX_TAGS_COMPACT) {
        if (!try_parse_tags(*this, p, n, format_hint)) throw Error(PGX_ERR_FORMAT, "tag file does not parse in the requested format");
        return;
    }
    // AUTO: the format that consumes the file exactly wins (they cannot both: the first 8 bytes
    // are a byte count in one and a bit count in the other).
    if (try_parse_tags(*this, p, n, PGX_TAGS_COMPACT)) return;
    if (try_parse_tags(*this, p, n, PGX_TAGS_BYTECODE)) return;
    throw Error(PGX_ERR_FORMAT, "tag file is neither the ByteCode nor the sdsl-compact layout");
}

// ------------------------------------------------------------------------------------------
// Extension tables.  For a byte `a` the reference computes (src/r-index.cpp:713-756):
//   comp_idx = sym_map[complement(a)];  for b = 0.. while sym_map[nuc[b]] < comp_idx:
//       k' += R[sym_map[complement(nuc[b])]] - Q[...]
//   s = R[sym_map[a]] - Q[sym_map[a]];  k = Q[sym_map[a]] + C[sym_map[a]]
// where R/Q = rank_at_cached(k+s / k), whose slot i holds (encoded, :634-638) counts6[sym_map[nuc[i]]]
// or (legacy, :596-601) rank(nuc[i]) + cum[sym_map[nuc[i]]].  Everything above is a function of the
// byte alone, so it is tabulated here once; the kernels only see (code, slot, multiplicities).
static void complement_table(uint8_t comp[256]) { // src/r-index.cpp:1512-1529
    for (int i = 0; i < 256; i++) comp[i] = (uint8_t)i;
    comp['A'] = 'T'; comp['C'] = 'G'; comp['G'] = 'C'; comp['T'] = 'A';
    comp['a'] = 't'; comp['c'] = 'g'; comp['g'] = 'c'; comp['t'] = 'a';
}

void build_ext_tables(const RiFile &ri, uint32_t mode, PgxConsts &c) {
    const uint32_t sigma = (uint32_t)ri.C.size();
    uint8_t comp[256];
    complement_table(comp);
    c.sigma = sigma;
    for (int i = 0; i < 8; i++) { c.C[i] = i < (int)sigma ? ri.C[i] : 0; c.slot_code[i] = 0; }
    c.excl_mask = 0;
    if (mode == PGX_MODE_STRICT) {
        static const int COMP_CODE[6] = {0, 5, 3, 2, 4, 1};
        for (uint32_t i = 0; i < sigma; i++) { // slot -> code view for pgx_rank_batch
            for (int code = 0; code < 6; code++)
                if ((code == 0 || ri.sym_map[kNuc[code]] != 0) && ri.sym_map[kNuc[code]] == i) c.slot_code[i] = (uint32_t)code;
        }
        for (int dir = 0; dir < 2; dir++)
            for (int byte = 0; byte < 256; byte++) {
                uint8_t a = dir ? comp[byte] : (uint8_t)byte;
                int code = code_of_byte(a);
                bool ok = code >= 0 && (code == 0 || ri.sym_map[a] != 0);
                uint32_t m = 0;
                if (ok)
                    for (int x = 0; x < COMP_CODE[code]; x++) m += 1u << (3 * COMP_CODE[x]);
                c.ext_tab[dir * 256 + byte] = ok ? PGX_EXT_MAKE(code, ri.sym_map[a], m, 0) : PGX_EXT_MAKE(0, 0, 0, 1);
            }
        return;
    }
    // COMPAT: slot i of the rank cache
    uint32_t slot_src[8] = {0};
    for (uint32_t i = 0; i < sigma; i++) {
        if (ri.encoded) {
            slot_src[i] = ri.sym_map[kNuc[i]]; // counts6[sym_map[nuc[i]]]
        } else {
            bool present = (i == 0) || ri.sym_map[kNuc[i]] != 0;
            slot_src[i] = i; // absent symbol: header count of code i holds cum[0] of the reference block
            if (!present) c.excl_mask |= 1u << i;
        }
        c.slot_code[i] = slot_src[i];
    }
    for (int dir = 0; dir < 2; dir++)
        for (int byte = 0; byte < 256; byte++) {
            uint8_t a = dir ? comp[byte] : (uint8_t)byte; // forward_extend: backward by complement (:761)
            uint32_t v = ri.sym_map[a];
            uint32_t comp_idx = ri.sym_map[comp[a]];
            uint32_t mult[6] = {0, 0, 0, 0, 0, 0};
            for (int b = 0; b < 6 && ri.sym_map[kNuc[b]] < comp_idx; b++) {
                uint32_t idx = ri.sym_map[comp[kNuc[b]]];
                mult[slot_src[idx]]++;
            }
            uint32_t m = 0;
            for (int code = 0; code < 6; code++) m |= mult[code] << (3 * code);
            c.ext_tab[dir * 256 + byte] = PGX_EXT_MAKE(slot_src[v], v, m, 0);
        }
}

// Unidirectional backward search tables (query_tags path).
//   legacy  FastLocate::count -> LF (src/r-index.cpp:650-687): symbols with sym_map == 0 are rejected
//           (:653); rankAt counts runs whose byte equals the symbol (r-index.hpp:180-221)
//   encoded count_encoded -> LF_encoded (:689-711) -> rankAt_encoded (:570-590): no rejection, the ranked
//           code is symbol_to_code(sym) (unknown -> 0).  rankAt_encoded always parses SIX cumulative
//           varints (quirk 3): on an encoded index without N the reference mis-parses every block, which
//           a rank image cannot (and should not) reproduce -> count_supported = 0 in COMPAT.
static void build_count_table(const RiFile &ri, uint32_t mode, PgxConsts &c) {
    c.count_supported = 1;
    for (int byte = 0; byte < 256; byte++) {
        const int code = code_of_byte((uint8_t)byte);
        const uint32_t v = ri.sym_map[byte];
        uint32_t e;
        if (mode == PGX_MODE_STRICT) {
            const bool ok = code >= 1 && v != 0;
            e = ok ? PGX_EXT_MAKE(code, v, 0, 0) : PGX_EXT_MAKE(0, 0, 0, 1);
        } else if (!ri.encoded) {
            e = v != 0 ? PGX_EXT_MAKE(code, v, 0, 0) : PGX_EXT_MAKE(0, 0, 0, 1);
        } else {
            e = PGX_EXT_MAKE(code >= 0 ? code : 0, v, 0, 0);
        }
        c.cnt_tab[byte] = e;
    }
    if (mode != PGX_MODE_STRICT && ri.encoded && ri.C.size() != 6) c.count_supported = 0;
}

// ------------------------------------------------------------------------------------------
static void put_block(std::vector<uint8_t> &blocks, const uint64_t c6[6], const std::vector<std::pair<uint8_t, uint32_t>> &ent) {
    uint32_t dw[16] = {0};
    for (int i = 0; i < 6; i++) dw[i] = (uint32_t)c6[i];
    for (int i = 0; i < 4; i++) dw[6] |= (uint32_t)((c6[i] >> 32) & 0xFF) << (8 * i);
    dw[7] = (uint32_t)((c6[4] >> 32) & 0xFF) | ((uint32_t)((c6[5] >> 32) & 0xFF) << 8) | ((uint32_t)ent.size() << 16);
    for (size_t e = 0; e < ent.size(); e++) {
        uint32_t v = ((3u * (uint32_t)ent[e].first) << PGX_RUN_LEN_BITS) | ent[e].second;
        dw[8 + e / 2] |= v << (16 * (e & 1));
    }
    const uint8_t *p = reinterpret_cast<const uint8_t *>(dw);
    blocks.insert(blocks.end(), p, p + PGX_BLOCK_BYTES);
}

// literal count image (quirk 3; pgx_device.h PgxLitImage): the reference's own 10-run blocks with their true cumulative
// counts (nuc order; N forced to 0, src/r-index.cpp:80) and the runs its late-starting scan sees
void build_literal_image(const RiFile &ri, LitHostImage &m) {
    if (m.built) return;
    const uint64_t nb = ri.lit_runs.size();
    m.bstart = ri.blocks_start_pos.ones;
    m.cum.assign(nb * 6, 0);
    m.roff.assign(nb + 1, 0);
    m.runs.clear();
    static const int slot_of_nuc[6] = {0, 1, 2, 3, -1, 4}; // \n A C G (N) T in sym_map order of a no-N index
    for (uint64_t b = 0; b < nb; b++) {
        for (int i = 0; i < 6; i++)
            if (slot_of_nuc[i] >= 0 && (size_t)slot_of_nuc[i] < ri.blocks[b].cum.size()) m.cum[b * 6 + i] = ri.blocks[b].cum[slot_of_nuc[i]];
        m.runs.insert(m.runs.end(), ri.lit_runs[b].begin(), ri.lit_runs[b].end());
        if (m.runs.size() >> 32) throw Error(PGX_ERR_UNSUPPORTED, "literal count image: too many runs");
        m.roff[b + 1] = (uint32_t)m.runs.size();
    }
    for (int c = 0; c < 256; c++) {
        const int code = code_of_byte((uint8_t)c);
        m.code_of[c] = code < 0 ? 0u : (uint32_t)code; // symbol_to_code (r-index.hpp:664-668): unknown -> 0
        m.cslot_of[c] = ri.sym_map[c];
    }
    for (int i = 0; i < 8; i++) m.C[i] = i < (int)ri.C.size() ? ri.C[i] : 0;
    if (m.runs.empty()) m.runs.push_back(0);
    m.built = true;
}

// dense image: 64 symbols per 64-byte block as three bit planes under the usual count header (pgx_image.h)
static void build_dense_image(const RiFile &ri, HostImage &img) {
    PgxConsts &c = img.consts;
    const uint64_t nb = (c.n >> 6) + 1;
    if (nb >> 32) throw Error(PGX_ERR_UNSUPPORTED, "BWT too long for the dense image");
    img.blocks.assign(nb * PGX_BLOCK_BYTES, 0);
    img.bstart.clear();
    img.dir.assign(1, 0);
    img.blow.clear();
    uint32_t *dw = reinterpret_cast<uint32_t *>(img.blocks.data());
    uint64_t c6[6] = {0, 0, 0, 0, 0, 0};
    auto put_header = [&](uint64_t b) {
        uint32_t *h = dw + b * 16;
        for (int i = 0; i < 6; i++) h[i] = (uint32_t)c6[i];
        for (int i = 0; i < 4; i++) h[6] |= (uint32_t)((c6[i] >> 32) & 0xFF) << (8 * i);
        h[7] = (uint32_t)((c6[4] >> 32) & 0xFF) | ((uint32_t)((c6[5] >> 32) & 0xFF) << 8);
    };
    uint64_t pos = 0, n_runs = 0;
    put_header(0);
    for (const auto &blk : ri.blocks)
        for (const auto &ru : blk.runs) {
            n_runs++;
            const uint32_t code = ru.first;
            uint64_t left = ru.second;
            while (left) {
                const uint64_t b = pos >> 6;
                const uint32_t off = (uint32_t)(pos & 63);
                const uint64_t take = std::min<uint64_t>(left, 64 - off);
                const uint64_t bits = (take == 64 ? ~0ull : ((1ull << take) - 1ull)) << off;
                uint32_t *h = dw + b * 16;
                for (int p = 0; p < 3; p++)
                    if ((code >> p) & 1u) {
                        h[8 + 2 * p] |= (uint32_t)bits;
                        h[9 + 2 * p] |= (uint32_t)(bits >> 32);
                    }
                c6[code] += take;
                left -= take;
                pos += take;
                if ((pos & 63) == 0) put_header(pos >> 6);
            }
        }
    if (pos != c.n) throw Error(PGX_ERR_FORMAT, "FastLocate: run lengths do not add up to the BWT size");
    img.n_runs = n_runs;
    c.n_blocks = (uint32_t)nb;
    c.dir_shift = 0;
    c.dir_entries = 1;
    c.image_kind = PGX_IMAGE_DENSE;
}

// dense2 image: 384 symbols per 128-byte block as three sub-blocks of two bit planes + exception runs for \n and N (pgx_image.h)
// (wide: header counts are deltas against the bases of the block's superblock, 2^sb_shift blocks each)
static uint32_t sb_shift_for(uint64_t nb, uint32_t want) { // the requested shift, raised until PGX_SB_MAX superblocks suffice
    uint32_t sh = want;
    while (((nb - 1) >> sh) + 1 > PGX_SB_MAX) sh++;
    return sh;
}
static void build_dense2_image(const RiFile &ri, HostImage &img, bool wide, uint32_t sb_shift_want) {
    PgxConsts &c = img.consts;
    if ((c.n >> 32) && !wide) throw Error(PGX_ERR_UNSUPPORTED, "BWT too long for the narrow dense2 image");
    const uint64_t nb = c.n / PGX_D2_SYMS + 1;
    if (nb >> 32) throw Error(PGX_ERR_UNSUPPORTED, "BWT too long for the dense2 image");
    const uint32_t sb_shift = wide ? sb_shift_for(nb, sb_shift_want) : 31u;
    if (wide && (PGX_D2_SYMS << sb_shift) > (1ull << 31)) throw Error(PGX_ERR_UNSUPPORTED, "BWT too long for the wide dense2 image (more than PGX_SB_MAX superblocks)");
    const uint64_t n_sb = wide ? ((nb - 1) >> sb_shift) + 1 : 1;
    img.sbase2.assign(n_sb * 8, 0);
    uint64_t base[6] = {0, 0, 0, 0, 0, 0};
    img.blocks.assign(nb * PGX_D2_BLOCK_BYTES, 0);
    img.bstart.clear();
    img.dir.assign(1, 0);
    img.blow.clear();
    img.exc.clear();
    uint32_t *dw = reinterpret_cast<uint32_t *>(img.blocks.data());
    uint64_t c6[6] = {0, 0, 0, 0, 0, 0};
    static const int hdr_slot[6] = {-1, 0, 1, 2, 4, 3}; // \n A C G N T -> header dword (A C G T N; \n is implied)
    static const int two_bit[6] = {0, 0, 1, 2, 0, 3};   // plane code (\n and N are exceptions stored as 0)
    auto put_header = [&](uint64_t b) {
        uint32_t *h = dw + b * 32;
        if (wide && (b & ((1ull << sb_shift) - 1)) == 0) { // a superblock starts here
            uint64_t *sb = img.sbase2.data() + (b >> sb_shift) * 8;
            uint64_t sum = 0;
            for (int i = 1; i < 6; i++) { base[i] = c6[i]; sb[hdr_slot[i]] = c6[i]; sum += c6[i]; }
            sb[5] = sum;
        }
        for (int i = 1; i < 6; i++) h[hdr_slot[i]] = (uint32_t)(c6[i] - base[i]);
        if (img.exc.size() >> 24) throw Error(PGX_ERR_UNSUPPORTED, "too many exception runs for the dense2 image");
        h[5] = (uint32_t)img.exc.size();
    };
    auto set_bits = [](uint32_t *words, uint32_t lo, uint32_t hi) { // bits [lo, hi) of a plane of 128 bits
        for (uint32_t w = lo >> 5; w <= ((hi - 1) >> 5); w++) {
            const uint32_t a = std::max(lo, w << 5) & 31u, e = std::min(hi, (w + 1) << 5) - (w << 5); // bits [a, e) of word w
            words[w] |= (e == 32 ? 0xFFFFFFFFu : ((1u << e) - 1u)) & ~((1u << a) - 1u);
        }
    };
    auto finish_block = [&](uint64_t b) { // sub-block counts from the finished planes
        uint32_t *h = dw + b * 32;
        uint64_t sub = 0;
        uint32_t n0 = 0, n1 = 0, n3 = 0;
        for (int s = 0; s < 2; s++) {
            for (int k = 0; k < 4; k++) {
                const uint32_t a = h[8 + 8 * s + k], d = h[12 + 8 * s + k];
                n0 += (uint32_t)__builtin_popcount(a); n1 += (uint32_t)__builtin_popcount(d); n3 += (uint32_t)__builtin_popcount(a & d);
            }
            sub |= ((uint64_t)n0 | ((uint64_t)n1 << 9) | ((uint64_t)n3 << 18)) << (27 * s);
        }
        h[6] = (uint32_t)sub; h[7] = (uint32_t)(sub >> 32);
    };
    uint64_t pos = 0, n_runs = 0;
    put_header(0);
    for (const auto &blk : ri.blocks)
        for (const auto &ru : blk.runs) {
            n_runs++;
            const uint32_t code = ru.first;
            uint64_t left = ru.second;
            while (left) {
                const uint64_t b = pos / PGX_D2_SYMS;
                const uint32_t off = (uint32_t)(pos % PGX_D2_SYMS);
                const uint64_t take = std::min<uint64_t>(left, PGX_D2_SYMS - off);
                uint32_t *h = dw + b * 32;
                const int tb = two_bit[code];
                if (tb)
                    for (uint32_t i = off; i < off + (uint32_t)take;) { // sub-block by sub-block: 4 dwords of plane 0, then 4 of plane 1
                        uint32_t *sb = h + 8 + 8 * (i >> 7);
                        const uint32_t lo = i & 127u, hi = std::min<uint32_t>(128u, lo + (off + (uint32_t)take - i));
                        if (tb & 1) set_bits(sb, lo, hi);
                        if (tb & 2) set_bits(sb + 4, lo, hi);
                        i += hi - lo;
                    }
                if (code == 0 || code == 4) {
                    const uint32_t kind = code == 4 ? 1u : 0u;
                    // extend the block's last exception run when this one continues it
                    if ((h[5] >> 24) && (img.exc.back() >> 18) == kind && (img.exc.back() & 511u) + ((img.exc.back() >> 9) & 511u) == off)
                        img.exc.back() += (uint32_t)take << 9;
                    else {
                        img.exc.push_back(off | ((uint32_t)take << 9) | (kind << 18));
                        h[5] += 1u << 24; // at most 192 alternating runs in 384 symbols
                    }
                }
                c6[code] += take;
                left -= take;
                pos += take;
                if (pos % PGX_D2_SYMS == 0) { finish_block(b); put_header(pos / PGX_D2_SYMS); }
            }
        }
    if (pos != c.n) throw Error(PGX_ERR_FORMAT, "FastLocate: run lengths do not add up to the BWT size");
    finish_block(nb - 1);
    if (img.exc.empty()) img.exc.push_back(0); // never indexed; keeps the device buffer non-empty
    img.n_runs = n_runs;
    c.n_blocks = (uint32_t)nb;
    c.dir_shift = 0;
    c.dir_entries = 1;
    c.image_kind = PGX_IMAGE_DENSE2;
    c.wide = wide ? 1u : 0u;
    c.d2_sb_shift = sb_shift;
    c.n_sb2 = (uint32_t)n_sb;
}

// PAIRS image (pgx_image.h): c2(p) = BWT[LF(p)] for every position whose BWT symbol is A C G T, pair counts per 128 positions.
// Returns false (and leaves img without one) when the index does not qualify: an extension entry that ranks a regular symbol must
// place its interval at that symbol's true C value -- the second step of a pair relies on LF mapping the positions with c1 = a
// onto the interval after the first -- and the special runs must fit ptab.
// stride: positions between block starts, PGX_PAIRS_SYMS (96: the blocks tile the BWT) or PGX_PAIRS_STRIDE64 (64: they overlap by 32).
// A block always covers PGX_PAIRS_SYMS positions.
static bool build_pairs_image(const RiFile &ri, HostImage &img, bool wide, uint32_t sb_shift_want, uint32_t stride) {
    if (stride != PGX_PAIRS_SYMS && stride != PGX_PAIRS_STRIDE64) return false;
    PgxConsts &c = img.consts;
    img.pairs.clear(); img.pbase.clear();
    c.has_pairs = 0; c.pair_runs = 0;
    const uint64_t n = c.n;
    if (n == 0 || ((n >> 32) && !wide) || c.excl_mask) return false;
    if (!wide) for (int i = 0; i < 8; i++) if (c.C[i] >> 32) return false;
    // the BWT as nuc codes, counts, true C
    std::vector<uint8_t> bw(n);
    uint64_t tot[6] = {0, 0, 0, 0, 0, 0};
    {
        uint64_t pos = 0;
        for (const auto &blk : ri.blocks)
            for (const auto &ru : blk.runs) {
                if (ru.first > 5 || pos + ru.second > n) throw Error(PGX_ERR_FORMAT, "FastLocate: run lengths do not add up to the BWT size");
                std::memset(bw.data() + pos, (int)ru.first, ru.second);
                tot[ru.first] += ru.second;
                pos += ru.second;
            }
        if (pos != n) throw Error(PGX_ERR_FORMAT, "FastLocate: run lengths do not add up to the BWT size");
    }
    uint64_t trueC[7] = {0};
    for (int i = 0; i < 6; i++) trueC[i + 1] = trueC[i] + tot[i];
    static const int two_bit[6] = {-1, 0, 1, 2, -1, 3}; // \n A C G N T
    static const int code_of_two[4] = {1, 2, 3, 5};
    for (int e = 0; e < 512; e++) {
        const uint32_t en = c.ext_tab[e];
        if (PGX_EXT_KILL(en)) continue;
        const uint32_t cv = PGX_EXT_CV(en);
        if (cv > 5) return false;
        if (two_bit[cv] < 0) continue;
        if (c.C[PGX_EXT_V(en)] != trueC[cv]) return false;
        // the other coordinate moves by the occurrences of the regular symbols that sort after this one (weight 1 each): the kernel counts
        // "first symbol > a" in one popcount chain instead of weighting four counts
        for (int y = 0; y < 4; y++)
            if (((PGX_EXT_M(en) >> (3 * code_of_two[y])) & 7u) != (y > two_bit[cv] ? 1u : 0u)) return false;
    }
    // chunks of whole blocks; per chunk the counts of the six codes (for LF), later of the sixteen pairs
    const uint64_t nb = n / (uint64_t)stride + 1;
    struct Special { uint64_t start, len; uint64_t cnt[5]; }; // cnt: {positions, c2 special with c1 = A, C, G, T}
    std::vector<std::vector<Special>> spec_of;
    std::vector<std::array<uint64_t, 6>> code_cnt;
    std::vector<std::array<uint64_t, 16>> pair_cnt;
    std::vector<std::pair<uint64_t, uint64_t>> range_of;
    {
        const unsigned hw = std::thread::hardware_concurrency();
        uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(hw ? hw : 1, 32), nb / 4096 + 1));
        if (const char *e = std::getenv("PGX_BUILD_THREADS")) want = std::max<uint64_t>(1, std::min<uint64_t>(std::strtoull(e, nullptr, 10), std::min<uint64_t>(nb, 256)));
        for (uint64_t t = 0; t < want; t++) range_of.push_back({nb * t / want, nb * (t + 1) / want});
    }
    const size_t nt = range_of.size();
    code_cnt.assign(nt, {}); pair_cnt.assign(nt, {}); spec_of.assign(nt, {});
    auto run_chunks = [&](auto &&fn) {
        std::vector<std::thread> th;
        std::exception_ptr err;
        std::mutex mu;
        for (size_t t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                try { fn(t, std::min<uint64_t>(range_of[t].first * (uint64_t)stride, n), std::min<uint64_t>(range_of[t].second * (uint64_t)stride, n)); }
                catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); }
            });
        for (auto &x : th) x.join();
        if (err) std::rethrow_exception(err);
    };
    run_chunks([&](size_t t, uint64_t p0, uint64_t p1) {
        std::array<uint64_t, 6> k{};
        for (uint64_t p = p0; p < p1; p++) k[bw[p]]++;
        code_cnt[t] = k;
    });
    // pr[p]: 0..15 = 4 c1 + c2 (2-bit codes); 0x80 = c1 special; 0x81 + y = c2 special, c1 = y
    std::vector<uint8_t> pr(n);
    run_chunks([&](size_t t, uint64_t p0, uint64_t p1) {
        uint64_t at[6];
        for (int i = 0; i < 6; i++) { at[i] = trueC[i]; for (size_t u = 0; u < t; u++) at[i] += code_cnt[u][i]; }
        std::array<uint64_t, 16> pc{};
        std::vector<Special> &sp = spec_of[t];
        for (uint64_t p = p0; p < p1; p++) {
            const int y = two_bit[bw[p]];
            uint8_t v;
            if (y < 0) v = 0x80;
            else {
                const int x = two_bit[bw[at[bw[p]]++]]; // BWT[LF(p)]
                v = x < 0 ? (uint8_t)(0x81 + y) : (uint8_t)(4 * y + x);
            }
            pr[p] = v;
            if (v & 0x80) {
                if (sp.empty() || sp.back().start + sp.back().len != p) sp.push_back({p, 0, {0, 0, 0, 0, 0}});
                sp.back().len++; sp.back().cnt[0]++;
                if (v > 0x80) sp.back().cnt[v - 0x80]++;
            } else pc[v]++;
        }
        pair_cnt[t] = pc;
    });
    // special runs in order (runs that meet at a chunk border are one)
    std::vector<Special> runs;
    for (size_t t = 0; t < nt; t++)
        for (const Special &sr : spec_of[t]) {
            if (!runs.empty() && runs.back().start + runs.back().len == sr.start) {
                runs.back().len += sr.len;
                for (int i = 0; i < 5; i++) runs.back().cnt[i] += sr.cnt[i];
            } else runs.push_back(sr);
        }
    {
        uint64_t special = 0;
        for (const Special &sr : runs) special += sr.cnt[0];
        if (special >> 31) return false; // (the header counts of special positions are 31-bit)
        uint64_t half[4] = {0, 0, 0, 0}; // positions with c2 special and c1 = A, C, G, T: 24-bit header fields (pgx_image.h)
        for (const Special &sr : runs) for (int i = 0; i < 4; i++) half[i] += sr.cnt[1 + i];
        for (int i = 0; i < 4; i++) if (half[i] >> 24) return false;
    }
    const bool with_ext = !(std::getenv("PGX_PAIRS_EXT") && std::getenv("PGX_PAIRS_EXT")[0] == '0');
    // positions with c2 special and c1 = A, C, G, T in every chunk (the blocks carry their running sums)
    std::vector<std::array<uint64_t, 4>> spec_cnt(nt, std::array<uint64_t, 4>{});
    for (size_t t = 0; t < nt; t++)
        for (const Special &sr : spec_of[t])
            for (int i = 0; i < 4; i++) spec_cnt[t][i] += sr.cnt[1 + i];
    // superblocks (wide): bases of the sixteen pair counts and their four row sums
    if ((nb >> 32)) return false;
    const uint32_t sb_shift = wide ? sb_shift_for(nb, sb_shift_want) : 31u;
    if (wide && ((uint64_t)(uint64_t)stride << sb_shift) > (1ull << 31)) return false;
    const uint64_t n_sb = wide ? ((nb - 1) >> sb_shift) + 1 : 1;
    img.pbase.assign(n_sb * 24, 0);
    if (wide) { // pair counts before every superblock: those before its chunk + a scan of the chunk's positions before it
        std::vector<std::thread> th;
        const unsigned par = (unsigned)std::min<uint64_t>(n_sb, 16);
        for (unsigned w = 0; w < par; w++)
            th.emplace_back([&, w]() {
                for (uint64_t sbi = w; sbi < n_sb; sbi += par) {
                    const uint64_t b = sbi << sb_shift;
                    size_t t = 0;
                    while (t + 1 < nt && range_of[t].second <= b) t++;
                    std::array<uint64_t, 16> pc{};
                    for (size_t u = 0; u < t; u++) for (int i = 0; i < 16; i++) pc[i] += pair_cnt[u][i];
                    for (uint64_t p = std::min<uint64_t>(range_of[t].first * (uint64_t)stride, n); p < std::min<uint64_t>(b * (uint64_t)stride, n); p++)
                        if (!(pr[p] & 0x80)) pc[pr[p]]++;
                    uint64_t *sb = img.pbase.data() + sbi * 24;
                    for (int i = 0; i < 16; i++) { sb[i] = pc[i]; sb[16 + (i >> 2)] += pc[i]; }
                }
            });
        for (auto &x : th) x.join();
    }
    // blocks
    img.pairs.assign(nb * PGX_PAIRS_BLOCK_BYTES, 0);
    uint32_t *dw = reinterpret_cast<uint32_t *>(img.pairs.data());
    run_chunks([&](size_t t, uint64_t p0, uint64_t p1) {
        std::array<uint64_t, 16> pc{};
        std::array<uint64_t, 4> ps{};
        for (size_t u = 0; u < t; u++) { for (int i = 0; i < 16; i++) pc[i] += pair_cnt[u][i]; for (int i = 0; i < 4; i++) ps[i] += spec_cnt[u][i]; }
        const uint64_t b0 = range_of[t].first, b1 = range_of[t].second;
        size_t r = std::lower_bound(runs.begin(), runs.end(), b0 * (uint64_t)stride, [](const Special &a, uint64_t v) { return a.start < v; }) - runs.begin(); // runs starting before the block
        for (uint64_t b = b0; b < b1; b++) {
            uint32_t *h = dw + b * 32;
            const uint64_t s0 = b * (uint64_t)stride, s1 = std::min<uint64_t>(s0 + PGX_PAIRS_SYMS, n);
            while (r < runs.size() && runs[r].start < s0) r++;
            const uint64_t *sb = img.pbase.data() + (wide ? (b >> sb_shift) : 0) * 24; // (zero in a narrow image)
            for (int i = 0; i < 16; i++) h[i] = (uint32_t)(pc[i] - sb[i]);
            const bool flag = (r > 0 && runs[r - 1].start + runs[r - 1].len > s0) || (r < runs.size() && runs[r].start < s1);
            for (int i = 0; i < 4; i++) h[16 + i] = (uint32_t)ps[i];
            if (flag) h[16] |= 0x80000000u;
            const uint64_t adv = std::min<uint64_t>(s0 + stride, n); // the running counts move on by the stride; the planes cover the whole block
            for (uint64_t p = s0; p < s1; p++) {
                const uint8_t v = pr[p];
                if (v & 0x80) { if (v > 0x80 && p < adv) ps[v - 0x81]++; continue; }
                const uint32_t i = (uint32_t)(p - s0), bit = 1u << (i & 31), w = i >> 5;
                if (v & 4) h[20 + w] |= bit;
                if (v & 8) h[23 + w] |= bit;
                if (v & 1) h[26 + w] |= bit;
                if (v & 2) h[29 + w] |= bit;
                if (p < adv) pc[v]++;
            }
            // run continuation (pgx_image.h): how far the pair / the first symbol of the block's last position goes on behind the block
            const uint64_t e0 = s0 + PGX_PAIRS_SYMS;
            if (with_ext && e0 < n && !(pr[e0 - 1] & 0x80)) {
                const uint8_t last = pr[e0 - 1];
                uint32_t xp = 0, x1 = 0;
                while (xp < 255u && e0 + xp < n && pr[e0 + xp] == last) xp++;
                auto c1_of = [](uint8_t v) { return v == 0x80 ? 0xFFu : (v > 0x80 ? (uint32_t)(v - 0x81) : (uint32_t)(v >> 2)); };
                while (x1 < 255u && e0 + x1 < n && c1_of(pr[e0 + x1]) == (uint32_t)(last >> 2)) x1++;
                h[17] |= xp << 24;
                h[18] |= x1 << 24;
            }
        }
        (void)p0; (void)p1;
    });
    // pair_t2[8 y + c]: number of c in BWT[0, true C of y)
    for (int y = 0; y < 4; y++) {
        const uint64_t P = trueC[code_of_two[y]];
        uint64_t cnt[6] = {0, 0, 0, 0, 0, 0};
        size_t t = 0;
        for (; t + 1 < nt && std::min<uint64_t>(range_of[t].second * (uint64_t)stride, n) <= P; t++)
            for (int i = 0; i < 6; i++) cnt[i] += code_cnt[t][i];
        for (uint64_t p = std::min<uint64_t>(range_of[t].first * (uint64_t)stride, n); p < P; p++) cnt[bw[p]]++;
        for (int i = 0; i < 6; i++) { c.pair_t2[8 * y + i] = (uint32_t)cnt[i]; c.pair_t2w[8 * y + i] = cnt[i]; }
    }
    c.has_pairs = 1;
    c.pair_runs = (uint32_t)runs.size();
    c.pairs_sb_shift = sb_shift;
    c.pairs_stride = stride;
    c.n_sbp = (uint32_t)n_sb;
    return true;
}

// layout of the device rank image: dense bit planes when that costs little memory or the run-length image would
// not stay cache resident either; PGX_MODE_IMAGE_* / the environment variable PGX_IMAGE force one
// returns PGX_IMAGE_RL / PGX_IMAGE_DENSE / PGX_IMAGE_DENSE2
// pairs: 0 = no PAIRS image, 1 = one if the index qualifies, 2 = required (forced)
// wide: the 64-bit form of dense2 / pairs (forced by PGX_MODE_IMAGE_WIDE / PGX_IMAGE=wide, automatic from 2^32 symbols on)
static uint32_t choose_image(const RiFile &ri, uint32_t mode_bits, const PgxConsts &c, int &pairs, bool &wide) {
    const bool can = c.excl_mask == 0;
    const uint32_t all = PGX_MODE_IMAGE_RL | PGX_MODE_IMAGE_DENSE | PGX_MODE_IMAGE_DENSE2 | PGX_MODE_IMAGE_PAIRS;
    uint32_t force = mode_bits & all;
    pairs = 0;
    wide = (mode_bits & PGX_MODE_IMAGE_WIDE) != 0;
    if (!force && !wide)
        if (const char *e = std::getenv("PGX_IMAGE")) {
            const std::string v(e);
            force = v == "dense" ? PGX_MODE_IMAGE_DENSE : v == "dense2" ? PGX_MODE_IMAGE_DENSE2 : v == "rl" ? PGX_MODE_IMAGE_RL : v == "pairs" ? PGX_MODE_IMAGE_PAIRS : 0u;
            if (v == "wide" || v == "wide-pairs") { wide = true; force = PGX_MODE_IMAGE_PAIRS; }
            if (v == "wide-dense2") { wide = true; force = PGX_MODE_IMAGE_DENSE2; }
        }
    if (force & (force - 1)) throw Error(PGX_ERR_ARG, "pgx_index_open: more than one image layout forced");
    if (wide && (force & (PGX_MODE_IMAGE_RL | PGX_MODE_IMAGE_DENSE))) throw Error(PGX_ERR_ARG, "pgx_index_open: PGX_MODE_IMAGE_WIDE goes with dense2 / pairs only");
    if (wide && !force) force = PGX_MODE_IMAGE_PAIRS;
    // dense2 blocks of 384 symbols in at most PGX_SB_MAX superblocks of 2^31 symbols
    const bool fits_wide = c.n < ((uint64_t)PGX_SB_MAX << 30);
    if (force & (PGX_MODE_IMAGE_DENSE | PGX_MODE_IMAGE_DENSE2 | PGX_MODE_IMAGE_PAIRS)) {
        if (!can) throw Error(PGX_ERR_UNSUPPORTED, "dense image: not available for a legacy-layout index without N in COMPAT mode");
        if ((force & (PGX_MODE_IMAGE_DENSE2 | PGX_MODE_IMAGE_PAIRS)) && (c.n >> 32)) {
            if (!fits_wide) throw Error(PGX_ERR_UNSUPPORTED, "dense2 image: BWT too long");
            wide = true;
        }
        if (force & PGX_MODE_IMAGE_PAIRS) pairs = wide && !(mode_bits & PGX_MODE_IMAGE_PAIRS) && !std::getenv("PGX_IMAGE") ? 1 : 2;
        return (force & PGX_MODE_IMAGE_DENSE) ? PGX_IMAGE_DENSE : PGX_IMAGE_DENSE2;
    }
    if ((force & PGX_MODE_IMAGE_RL) || !can) return PGX_IMAGE_RL;
    // tiny BWTs: the 64-byte layout, staged in LDS by the kernels (cheapest decode; pgx_runtime.hip: 80 padded bytes per block)
    const uint64_t dense_blocks = (c.n >> 6) + 1;
    if (dense_blocks * 80 + 16 <= 48 * 1024) return PGX_IMAGE_DENSE;
    // the 64-byte layout while it stays resident in the 256 MB memory-side cache (cheapest decode: n = 64 M, 1 M reads: 3.1 ms
    // against 4.4 ms with dense2); beyond that dense2 up to 2^32 symbols: n / 3 bytes, cache resident up to chromosome scale
    // (n = 640 M, 10 M reads: 30 ms against 36 ms with the 64-byte layout served from HBM)
    const uint64_t dense_bytes0 = dense_blocks * PGX_BLOCK_BYTES;
    // (with the two-step PAIRS image next to it where the index qualifies: searches behind the seed table run on that -- n = 64 M, 1 M reads:
    //  2.66 against 3.17 ms --, everything else on the 64-byte image)
    if (dense_bytes0 <= (224ull << 20)) { pairs = (c.n >> 32) ? 0 : 1; return PGX_IMAGE_DENSE; }
    if (!(c.n >> 32)) { pairs = 1; return PGX_IMAGE_DENSE2; } // + the two-step PAIRS image when the index qualifies
    if (fits_wide) { pairs = 1; wide = true; return PGX_IMAGE_DENSE2; } // the same in 64 bits (pgx_image.h "WIDE")
    uint64_t runs = 0;
    for (const auto &b : ri.blocks) runs += b.runs.size();
    const uint64_t dense_bytes = dense_blocks * PGX_BLOCK_BYTES, rl_bytes = (runs / PGX_BLOCK_RUNS + 1) * (PGX_BLOCK_BYTES + 4);
    const uint64_t cache = 192ull << 20; // what stays resident in the 256 MB memory-side cache next to reads and tags
    return (rl_bytes > cache && dense_bytes <= (64ull << 30)) ? PGX_IMAGE_DENSE : PGX_IMAGE_RL;
}

void build_rank_image(const RiFile &ri, uint32_t mode_bits, HostImage &img) {
    const uint32_t mode = mode_bits & PGX_MODE_MASK;
    PgxConsts &c = img.consts;
    std::memset(&c, 0, sizeof c);
    c.n = ri.sequence_size;
    c.mode = mode;
    build_ext_tables(ri, mode, c);
    build_count_table(ri, mode, c);
    int pairs = 0;
    bool wide = false;
    const uint32_t kind = choose_image(ri, mode_bits, c, pairs, wide);
    img.pairs.clear(); img.sbase2.clear(); img.pbase.clear();
    // superblock sizes (tests shrink them so that small indexes have many: PGX_SB_SHIFT = blocks of the dense2 image per superblock, log2)
    uint32_t sh2 = PGX_D2_SB_SHIFT, shp = PGX_PAIRS_SB_SHIFT;
    if (const char *e = std::getenv("PGX_SB_SHIFT")) { sh2 = (uint32_t)std::min<unsigned long>(std::strtoul(e, nullptr, 10), PGX_D2_SB_SHIFT); shp = sh2 + 2; }
    // PAIRS blocks every 96 positions (4 n / 3 bytes) or every 64 (2 n bytes: an interval of up to 32 positions never needs a second block, 6 % fewer
    // lines at chr22 scale): the overlapping form up to 64 GiB of image; PGX_PAIRS_STRIDE=96|64 forces one
    uint32_t pstride = (2 * c.n <= (64ull << 30)) ? PGX_PAIRS_STRIDE64 : PGX_PAIRS_SYMS;
    if (const char *e = std::getenv("PGX_PAIRS_STRIDE")) {
        const unsigned long v = std::strtoul(e, nullptr, 10);
        if (v == PGX_PAIRS_SYMS) pstride = PGX_PAIRS_SYMS;
        else if (v == PGX_PAIRS_STRIDE64) pstride = PGX_PAIRS_STRIDE64;
    }
    if (kind == PGX_IMAGE_DENSE) {
        build_dense_image(ri, img);
        if (pairs) (void)build_pairs_image(ri, img, false, shp, pstride);
        return;
    }
    if (kind == PGX_IMAGE_DENSE2) {
        build_dense2_image(ri, img, wide, sh2);
        if (pairs && !build_pairs_image(ri, img, wide, shp, pstride) && pairs == 2)
            throw Error(PGX_ERR_UNSUPPORTED, "pairs image: the index does not qualify (extension tables of a COMPAT quirk, or too many N / endmarker runs)");
        return;
    }
    // Device blocks must refine the reference's blocks only when a header slot carries the
    // reference-block cumulative endmarker count (legacy layout, absent symbol, COMPAT).
    const bool refine = c.excl_mask != 0;
    img.blocks.clear();
    img.bstart.clear();
    uint64_t c6[6] = {0, 0, 0, 0, 0, 0}; // true counts at the current position
    uint64_t head[6];
    std::vector<std::pair<uint8_t, uint32_t>> ent;
    uint64_t pos = 0, block_pos = 0, E = 0, n_runs = 0;
    auto open_block = [&]() {
        for (int i = 0; i < 6; i++) head[i] = (c.excl_mask >> i) & 1 ? E : c6[i];
        block_pos = pos;
        ent.clear();
    };
    auto close_block = [&]() {
        if (ent.empty()) return;
        put_block(img.blocks, head, ent);
        img.bstart.push_back(block_pos);
        ent.clear();
    };
    open_block();
    for (uint64_t b = 0; b < ri.blocks.size(); b++) {
        if (refine) {
            close_block();
            E = ri.blocks[b].cum[0]; // cum[sym_map[absent]] == cum[0] (src/r-index.cpp:600)
            open_block();
        }
        for (auto &ru : ri.blocks[b].runs) {
            n_runs++;
            uint64_t left = ru.second;
            while (left) {
                // merge into the previous entry of the same code when it still has room
                if (!ent.empty() && ent.back().first == ru.first && ent.back().second < PGX_RUN_LEN_MAX) {
                    uint32_t add = (uint32_t)std::min<uint64_t>(left, PGX_RUN_LEN_MAX - ent.back().second);
                    ent.back().second += add;
                    left -= add; pos += add; c6[ru.first] += add;
                    continue;
                }
                if (ent.size() == PGX_BLOCK_RUNS) { close_block(); open_block(); }
                uint32_t take = (uint32_t)std::min<uint64_t>(left, PGX_RUN_LEN_MAX);
                ent.emplace_back(ru.first, take);
                left -= take; pos += take; c6[ru.first] += take;
            }
        }
    }
    close_block();
    if (img.bstart.empty()) { // empty BWT: a single empty block keeps the kernels branch-free
        uint64_t z[6] = {0};
        put_block(img.blocks, z, {});
        img.bstart.push_back(0);
    }
    img.n_runs = n_runs;
    const uint64_t nb = img.bstart.size();
    if (nb >> 32) throw Error(PGX_ERR_UNSUPPORTED, "more than 2^32 device blocks");
    c.n_blocks = (uint32_t)nb;
    // directory: about one bucket per two blocks (measured best: fewer, hotter directory lines; 1 or 4 blocks
    // per bucket were 5 % / 18 % slower on the 64 M-symbol index); a 64-bit entry resolves buckets holding at most two
    // block starts by itself (one 8-byte load), denser buckets fall back to the 16-bit lows
    uint32_t shift = 0;
    while (shift < PGX_DIR_MAX_SHIFT && 2 * ((c.n >> (shift + 1)) + 2) >= nb) shift++;
    c.dir_shift = shift;
    c.dir_entries = (c.n >> shift) + 2;
    img.dir.resize(c.dir_entries);
    img.blow.resize(nb);
    const uint64_t mask = (1ull << shift) - 1;
    for (uint64_t b = 0; b < nb; b++) img.blow[b] = (uint16_t)(img.bstart[b] & mask);
    uint64_t bi = 0; // number of blocks with start < (i << shift)
    for (uint64_t i = 0; i < c.dir_entries; i++) {
        const uint64_t p = i << shift;
        while (bi < nb && img.bstart[bi] < p) bi++;
        uint64_t e = bi, cnt = 0; // blocks starting inside bucket i
        while (bi + cnt < nb && (img.bstart[bi + cnt] >> shift) == i) cnt++;
        e |= (cnt > 255 ? 255 : cnt) << 32;
        if (cnt >= 1) e |= (uint64_t)img.blow[bi] << 40;
        if (cnt >= 2) e |= (uint64_t)img.blow[bi + 1] << 52;
        img.dir[i] = e;
    }
}

void build_tag_image(const TagFile &tf, HostImage &img) {
    PgxConsts &c = img.consts;
    const uint64_t nr = tf.bwt_intervals.ones.size();
    img.tstart = tf.bwt_intervals.ones;
    img.tvals.resize(tf.items.size());
    for (uint64_t i = 0; i < tf.items.size(); i++) {
        uint64_t v = tf.items[i];
        // decode_run (src/tag_arrays.cpp:59-70, length_bits=9) / decode_run_length_compact (:47-55)
        // followed by gbwtgraph::Position::encode = node << 11 | rev << 10 | offset
        uint64_t off = v & 0x3FF, rev = (v >> 10) & 1;
        uint64_t node = tf.format == PGX_TAGS_BYTECODE ? v >> 20 : v >> 11;
        img.tvals[i] = (node << 11) | (rev << 10) | off;
    }
    c.has_tags = 1;
    c.n_tag_runs = nr;
    uint64_t span = tf.bwt_intervals.size ? tf.bwt_intervals.size : 1;
    uint32_t shift = 0;
    while (shift < 40 && ((span >> (shift + 1)) + 2) >= 2 * (nr + 1)) shift++;
    c.tag_dir_shift = shift;
    c.tag_dir_entries = (span >> shift) + 2;
    // tdir[i] = number of run starts <= (i << shift)   (an upper_bound seed)
    img.tdir.resize(c.tag_dir_entries);
    uint64_t k = 0;
    for (uint64_t i = 0; i < c.tag_dir_entries; i++) {
        uint64_t p = i << shift;
        while (k < nr && img.tstart[k] <= p) k++;
        img.tdir[i] = (uint32_t)k;
    }
    if (nr >> 32) throw Error(PGX_ERR_UNSUPPORTED, "more than 2^32 tag runs");
}

// sampled directory over an ascending array: dir[i] = #elements <= (i << shift), about one bucket per element
static void build_sorted_dir(const std::vector<uint64_t> &arr, uint64_t span, uint32_t &shift, uint64_t &entries,
                             std::vector<uint32_t> &dir) {
    const uint64_t cnt = arr.size();
    if (cnt >> 32) throw Error(PGX_ERR_UNSUPPORTED, "more than 2^32 runs");
    if (!span) span = 1;
    shift = 0;
    while (shift < 48 && ((span >> (shift + 1)) + 2) >= 2 * (cnt + 1)) shift++;
    entries = (span >> shift) + 2;
    dir.resize(entries);
    uint64_t k = 0;
    for (uint64_t i = 0; i < entries; i++) {
        const uint64_t p = i << shift;
        while (k < cnt && arr[k] <= p) k++;
        dir[i] = (uint32_t)k;
    }
}

// locate image: run starts in the reference's run numbering (the file's blocks as written: endmarker runs are one
// symbol each, src/r-index.cpp:842-900), head samples, and the tail -> next-head map of locateNext (:1363-1366)
void build_locate_image(const RiFile &ri, LocHostImage &loc) {
    if (loc.built) return;
    const uint64_t r = ri.samples.size();
    loc.rstart.clear();
    loc.rstart.reserve(r + 1);
    uint64_t pos = 0;
    for (size_t b = 0; b < ri.blocks.size(); b++)
        for (const auto &run : ri.blocks[b].runs) {
            loc.rstart.push_back(pos);
            pos += run.second;
        }
    if (loc.rstart.size() != r) throw Error(PGX_ERR_FORMAT, "FastLocate: samples and runs disagree (" + std::to_string(r) + " samples, " +
                                                       std::to_string(loc.rstart.size()) + " runs)");
    if (pos != ri.sequence_size) throw Error(PGX_ERR_FORMAT, "FastLocate: run lengths do not add up to the BWT size");
    loc.rstart.push_back(pos);
    loc.rsamp.resize(r);
    for (uint64_t i = 0; i < r; i++) loc.rsamp[i] = ri.samples.get(i);
    const uint64_t nl = ri.last.ones.size();
    if (ri.last_to_run.size() != nl) throw Error(PGX_ERR_FORMAT, "FastLocate: last and last_to_run disagree");
    loc.lpos = ri.last.ones;
    loc.lnext.resize(nl);
    for (uint64_t i = 0; i < nl; i++) {
        const uint64_t run = ri.last_to_run.get(i) + 1;
        loc.lnext[i] = run < r ? loc.rsamp[run] : PGX_NO_POSITION; // the reference reads samples[r] (past the end) here
    }
    PgxLocConsts &c = loc.consts;
    std::memset(&c, 0, sizeof c);
    c.n = ri.sequence_size;
    c.n_runs = r;
    c.n_last = nl;
    c.max_length = ri.max_length ? ri.max_length : 1;
    build_sorted_dir(loc.rstart, ri.sequence_size + 1, c.rdir_shift, c.rdir_entries, loc.rdir);
    build_sorted_dir(loc.lpos, ri.last.size + 1, c.ldir_shift, c.ldir_entries, loc.ldir);
    loc.built = true;
}

} // namespace pgx

// ------------------------------------------------------------------------------------------
// C ABI (host part)
using namespace pgx;

extern "C" const char *pgx_last_error(void) { return pgx::last_error().c_str(); }
extern "C" int pgx_abi_version(void) { return PGX_ABI_VERSION; }

#define PGX_GUARD_BEGIN try {
#define PGX_GUARD_END                                                    \
    }                                                                    \
    catch (const pgx::Error &e) { pgx::set_last_error(e.what()); return e.code; } \
    catch (const std::bad_alloc &) { pgx::set_last_error("out of host memory"); return PGX_ERR_NOMEM; } \
    catch (const std::exception &e) { pgx::set_last_error(e.what()); return PGX_ERR_FORMAT; }

void pgx_release_device_images(pgx_index *h); // pgx_runtime.hip

static pgx_index *open_impl(const uint8_t *ri, uint64_t ri_n, const uint8_t *tags, uint64_t tags_n, uint32_t tags_format, uint32_t mode) {
    if ((mode & PGX_MODE_MASK) > PGX_MODE_STRICT || (mode & ~(PGX_MODE_MASK | PGX_MODE_IMAGE_RL | PGX_MODE_IMAGE_DENSE | PGX_MODE_IMAGE_DENSE2 | PGX_MODE_IMAGE_PAIRS | PGX_MODE_IMAGE_WIDE)))
        throw Error(PGX_ERR_ARG, "pgx_index_open: bad mode");
    if (!ri && !tags) throw Error(PGX_ERR_ARG, "pgx_index_open: neither an r-index nor a tag array given");
    std::unique_ptr<pgx_index> h(new pgx_index());
    h->mode = mode & PGX_MODE_MASK;
    std::memset(&h->img.consts, 0, sizeof h->img.consts);
    if (ri) {
        h->ri.parse(ri, ri_n);
        build_rank_image(h->ri, mode, h->img);
        h->has_rank = true;
    }
    if (tags) {
        h->tags.parse(tags, tags_n, tags_format);
        build_tag_image(h->tags, h->img);
        h->has_tags = true;
    }
    return h.release();
}

extern "C" pgx_status pgx_index_open(const char *ri_path, const char *tags_path, uint32_t tags_format,
                                     uint32_t mode, pgx_index **out) {
    PGX_GUARD_BEGIN
    if (!ri_path || !out) throw Error(PGX_ERR_ARG, "pgx_index_open: null argument");
    if ((mode & PGX_MODE_MASK) > PGX_MODE_STRICT || (mode & ~(PGX_MODE_MASK | PGX_MODE_IMAGE_RL | PGX_MODE_IMAGE_DENSE | PGX_MODE_IMAGE_DENSE2 | PGX_MODE_IMAGE_PAIRS | PGX_MODE_IMAGE_WIDE)))
        throw Error(PGX_ERR_ARG, "pgx_index_open: bad mode");
    *out = nullptr;
    std::vector<uint8_t> f, t;
    try { f = read_whole_file(ri_path); }
    catch (const Error &) { throw Error(PGX_ERR_IO, std::string("Cannot open r-index: ") + ri_path); } // find_mems.cpp:30
    if (tags_path) t = read_whole_file(tags_path);
    *out = open_impl(f.data(), f.size(), tags_path ? t.data() : nullptr, t.size(), tags_format, mode);
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_index_open_memory(const void *ri_bytes, uint64_t ri_n, const void *tags_bytes, uint64_t tags_n,
                                            uint32_t tags_format, uint32_t mode, pgx_index **out) {
    PGX_GUARD_BEGIN
    if (!out) throw Error(PGX_ERR_ARG, "pgx_index_open_memory: null argument");
    *out = nullptr;
    *out = open_impl((const uint8_t *)ri_bytes, ri_n, (const uint8_t *)tags_bytes, tags_n, tags_format, mode);
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_index_tables(const pgx_index *h, uint8_t sym_map[256], uint64_t C[8], uint8_t complement[256]) {
    PGX_GUARD_BEGIN
    if (!h || !h->has_rank) throw Error(PGX_ERR_ARG, "pgx_index_tables: no r-index loaded");
    if (sym_map) std::memcpy(sym_map, h->ri.sym_map, 256);
    if (C) for (int i = 0; i < 8; i++) C[i] = i < (int)h->ri.C.size() ? h->ri.C[i] : 0;
    if (complement) complement_table(complement);
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" void pgx_index_close(pgx_index *h) {
    if (!h) return;
    pgx_release_device_images(h);
    delete h;
}

extern "C" pgx_status pgx_index_info_get(const pgx_index *h, pgx_index_info *info) {
    PGX_GUARD_BEGIN
    if (!h || !info) throw Error(PGX_ERR_ARG, "pgx_index_info_get: null argument");
    std::memset(info, 0, sizeof *info);
    const PgxConsts &c = h->img.consts;
    info->bwt_size = c.n;
    info->sigma = c.sigma;
    info->n_sequences = h->ri.C.size() > 1 ? h->ri.C[1] - h->ri.C[0] : 0; // tot_strings, r-index.hpp:484
    info->n_ref_blocks = h->ri.n_file_blocks;
    info->n_runs = h->img.n_runs;
    info->n_dev_blocks = c.n_blocks;
    info->dir_entries = c.dir_entries;
    info->dir_shift = c.dir_shift;
    info->is_encoded = h->ri.encoded;
    info->has_N = h->ri.hasN;
    info->mode = h->mode;
    info->has_tags = h->has_tags;
    info->tag_format = h->tags.format;
    info->n_tag_runs = c.n_tag_runs;
    info->tag_dir_entries = c.tag_dir_entries;
    info->tag_dir_shift = c.tag_dir_shift;
    info->image_bytes = h->img.blocks.size() + h->img.dir.size() * 8 + h->img.blow.size() * 2 + h->img.exc.size() * 4 + h->img.pairs.size() +
                        (h->img.sbase2.size() + h->img.pbase.size()) * 8;
    info->tag_image_bytes = h->img.tstart.size() * 8 + h->img.tvals.size() * 8 + h->img.tdir.size() * 4;
    info->image_in_lds = h->img.consts.image_kind == PGX_IMAGE_DENSE ? (uint64_t)h->img.consts.n_blocks * 80 + 16 <= 48 * 1024
                                                                     : (h->img.consts.image_kind == PGX_IMAGE_RL && info->image_bytes <= 48 * 1024);
    info->ref_block_mean_bytes = h->ri.ref_block_mean_bytes;
    info->max_length = h->ri.max_length;
    info->n_samples = h->ri.samples.size();
    info->image_kind = c.image_kind;
    info->image_pairs = c.has_pairs;
    info->image_wide = c.wide;
    info->pairs_stride = c.has_pairs ? c.pairs_stride : 0;
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_index_image_view(const pgx_index *h, int which, const void **ptr, uint64_t *bytes) {
    PGX_GUARD_BEGIN
    if (!h || !ptr || !bytes) throw Error(PGX_ERR_ARG, "pgx_index_image_view: null argument");
    const HostImage &m = h->img;
    if (which >= 8 && which <= 14) { // locate image (built on first use)
        if (!h->has_rank) throw Error(PGX_ERR_ARG, "pgx_index_image_view: no r-index loaded");
        build_locate_image(h->ri, const_cast<pgx_index *>(h)->loc);
    }
    const LocHostImage &l = h->loc;
    switch (which) {
    case 8: *ptr = l.rstart.data(); *bytes = l.rstart.size() * 8; return PGX_OK;
    case 9: *ptr = l.rsamp.data(); *bytes = l.rsamp.size() * 8; return PGX_OK;
    case 10: *ptr = l.rdir.data(); *bytes = l.rdir.size() * 4; return PGX_OK;
    case 11: *ptr = l.lpos.data(); *bytes = l.lpos.size() * 8; return PGX_OK;
    case 12: *ptr = l.lnext.data(); *bytes = l.lnext.size() * 8; return PGX_OK;
    case 13: *ptr = l.ldir.data(); *bytes = l.ldir.size() * 4; return PGX_OK;
    case 14: *ptr = &l.consts; *bytes = sizeof(PgxLocConsts); return PGX_OK;
    default: break;
    }
    switch (which) {
    case 0: *ptr = m.blocks.data(); *bytes = m.blocks.size(); break;
    case 1: *ptr = m.dir.data(); *bytes = m.dir.size() * 8; break;
    case 2: *ptr = m.bstart.data(); *bytes = m.bstart.size() * 8; break;
    case 3: *ptr = m.tstart.data(); *bytes = m.tstart.size() * 8; break;
    case 4: *ptr = m.tvals.data(); *bytes = m.tvals.size() * 8; break;
    case 5: *ptr = m.tdir.data(); *bytes = m.tdir.size() * 4; break;
    case 6: *ptr = &m.consts; *bytes = sizeof(PgxConsts); break;
    case 7: *ptr = m.blow.data(); *bytes = m.blow.size() * 2; break;
    case 15: *ptr = m.exc.data(); *bytes = m.exc.size() * 4; break;
    case 20: *ptr = m.pairs.data(); *bytes = m.pairs.size(); break;
    case 22: *ptr = m.sbase2.data(); *bytes = m.sbase2.size() * 8; break;
    case 23: *ptr = m.pbase.data(); *bytes = m.pbase.size() * 8; break;
    case 16: case 17: case 18: case 19: {
        if (!(h->ri.encoded && !h->ri.hasN)) throw Error(PGX_ERR_ARG, "pgx_index_image_view: the literal count image exists for encoded indexes without N only");
        pgx::LitHostImage &l = const_cast<pgx_index *>(h)->lit;
        build_literal_image(h->ri, l);
        if (which == 16) { *ptr = l.bstart.data(); *bytes = l.bstart.size() * 8; }
        else if (which == 17) { *ptr = l.cum.data(); *bytes = l.cum.size() * 8; }
        else if (which == 18) { *ptr = l.runs.data(); *bytes = l.runs.size() * 8; }
        else { *ptr = l.roff.data(); *bytes = l.roff.size() * 4; }
        break;
    }
    default: throw Error(PGX_ERR_ARG, "pgx_index_image_view: unknown view");
    }
    return PGX_OK;
    PGX_GUARD_END
}

// ------------------------------------------------------------------------------------------
// Host-side packing of a batch of reads for pgx_batch_upload_packed: two bits per symbol, 16 symbols per word ((byte >> 1) & 3: A C T G =
// 0 1 2 3, symbol i of the concatenation in bits 2 (i & 15) of word i >> 4 -- what pgx_bad_chunks_kernel writes when the reads arrive as
// bytes), and the list of reads that hold any other byte, whose bytes travel as they are.
namespace {
// eight bytes -> eight codes in 16 bits; bad = nonzero where a byte is not one of A C G T (upper case)
inline uint32_t pack8(uint64_t x, uint64_t &bad) {
    const uint64_t c = (x >> 1) & 0x0303030303030303ull;
    const uint64_t b0 = c & 0x0101010101010101ull, b1 = (c >> 1) & 0x0101010101010101ull;
    const uint64_t recon = 0x4141414141414141ull + 2 * (b0 & ~b1) + 0x13 * (b1 & ~b0) + 6 * (b0 & b1);
    bad = x ^ recon;
    uint64_t t = (c | (c >> 6)) & 0x000F000F000F000Full;
    t = (t | (t >> 12)) & 0x000000FF000000FFull;
    t = (t | (t >> 24)) & 0xFFFFull;
    return (uint32_t)t;
}
} // namespace

extern "C" pgx_status pgx_pack_reads(const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads, uint32_t threads, uint32_t *packed,
                                     uint64_t *side_ids, uint64_t side_ids_cap, uint8_t *side_bytes, uint64_t side_bytes_cap, uint64_t *n_side,
                                     uint64_t *n_side_bytes) {
    PGX_GUARD_BEGIN
    if (!offsets || !n_side || !n_side_bytes || (n_reads && offsets[n_reads] != offsets[0] && (!reads || !packed)))
        throw Error(PGX_ERR_ARG, "pgx_pack_reads: null argument");
    const uint64_t lo = offsets[0], n_bytes = offsets[n_reads] - lo, n_chunks = (n_bytes + 15) >> 4;
    const uint8_t *src = reads + lo;
    if (threads == 0) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 2));
    threads = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(threads, n_chunks / 65536 + 1));
    std::vector<std::vector<uint64_t>> bad_chunks(threads); // chunks with a byte outside A C G T, per slice, ascending
    auto slice = [&](uint32_t t) {
        const uint64_t c0 = n_chunks * t / threads, c1 = n_chunks * (t + 1) / threads;
        for (uint64_t c = c0; c < c1; c++) {
            uint64_t x0 = 0, x1 = 0, b0, b1;
            const uint64_t at = c << 4, left = n_bytes - at;
            if (left >= 16) { std::memcpy(&x0, src + at, 8); std::memcpy(&x1, src + at + 8, 8); }
            else { uint8_t tmp[16] = {0}; std::memcpy(tmp, src + at, (size_t)left); std::memcpy(&x0, tmp, 8); std::memcpy(&x1, tmp + 8, 8); }
            const uint32_t p0 = pack8(x0, b0), p1 = pack8(x1, b1);
            packed[c] = p0 | (p1 << 16);
            if (left < 8) { b0 &= (1ull << (8 * left)) - 1ull; b1 = 0; }
            else if (left < 16) b1 &= (1ull << (8 * (left - 8))) - 1ull;
            if (b0 | b1) bad_chunks[t].push_back(c);
        }
    };
    if (threads == 1) slice(0);
    else {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < threads; t++) th.emplace_back(slice, t);
        for (auto &x : th) x.join();
    }
    // the reads those bytes belong to (ascending, each once): chunks and reads are both in order, one forward walk
    uint64_t ns = 0, nb = 0, rid = 0;
    bool overflow = false;
    uint64_t last = ~0ull;
    for (uint32_t t = 0; t < threads; t++)
        for (uint64_t c : bad_chunks[t]) {
            const uint64_t at = c << 4, end = std::min(n_bytes, at + 16);
            for (uint64_t p = at; p < end; p++) {
                const uint8_t ch = src[p];
                if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') continue;
                while (offsets[rid + 1] - lo <= p) rid++; // (p < n_bytes: some read holds it)
                if (rid == last) continue;
                last = rid;
                const uint64_t len = offsets[rid + 1] - offsets[rid];
                if (ns < side_ids_cap && nb + len <= side_bytes_cap && side_ids && side_bytes) {
                    side_ids[ns] = rid;
                    std::memcpy(side_bytes + nb, reads + offsets[rid], (size_t)len);
                } else overflow = true;
                ns++; nb += len;
            }
        }
    *n_side = ns; *n_side_bytes = nb; // (what the batch needs, whether or not it fitted)
    if (overflow) throw Error(PGX_ERR_NOMEM, "pgx_pack_reads: the side list does not fit the caller's buffers (upload the reads as bytes instead)");
    return PGX_OK;
    PGX_GUARD_END
}

// pgx_host.hpp -- host-side (C++17) internals of libpgx: file parsers, image builder, writers.
// Nothing here computes queries: rank / extend / find_mems / tag queries exist only as HIP kernels.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pgx.h"
#include "pgx_image.h"

namespace pgx {

struct Error : std::runtime_error {
    pgx_status code;
    Error(pgx_status c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string &m);

// nuc order of include/pangenome_index/utils.hpp:11
static const uint8_t kNuc[6] = {'\n', 'A', 'C', 'G', 'N', 'T'};
inline int code_of_byte(uint8_t c) {
    for (int i = 0; i < 6; i++)
        if (kNuc[i] == c) return i;
    return -1;
}

// ------------------------------------------------------------------------------------------
// Serialised SDSL containers (vgteam/sdsl-lite classic `serialize` layout)
struct ByteReader {
    const uint8_t *p;
    uint64_t n, o = 0;
    ByteReader(const uint8_t *p_, uint64_t n_) : p(p_), n(n_) {}
    void need(uint64_t k, const char *what) const {
        if (o + k > n || o + k < o) throw Error(PGX_ERR_FORMAT, std::string("truncated file while reading ") + what);
    }
    template <class T> T get(const char *what) {
        need(sizeof(T), what);
        T v;
        std::memcpy(&v, p + o, sizeof(T));
        o += sizeof(T);
        return v;
    }
};

struct IntVector { // int_vector<w>; width 0 in the type = stored width byte
    uint64_t bits = 0;
    uint8_t width = 64;
    std::vector<uint64_t> words;
    uint64_t size() const { return width ? bits / width : 0; }
    uint64_t get(uint64_t i) const {
        uint64_t bit = i * width, wd = bit >> 6, sh = bit & 63;
        uint64_t x = words[wd] >> sh;
        if (sh + width > 64) x |= words[wd + 1] << (64 - sh);
        return width == 64 ? x : (x & ((1ULL << width) - 1));
    }
    void read(ByteReader &r, int fixed_width, const char *what);
    void write(std::vector<uint8_t> &out, bool with_width_byte) const;
    static IntVector pack(const std::vector<uint64_t> &vals, uint8_t width);
};

struct SdVector { // sd_vector<>: decoded to the ascending list of set positions
    uint64_t size = 0;
    std::vector<uint64_t> ones;
    void read(ByteReader &r, const char *what);
    void write(std::vector<uint8_t> &out) const; // incl. select_support_mcl<1>/<0> on the high bits
};

uint64_t bytecode_read(const uint8_t *s, uint64_t n, uint64_t &i, const char *what);
void bytecode_write(std::vector<uint8_t> &out, uint64_t v);

// ------------------------------------------------------------------------------------------
struct RefBlock {                       // one 10-run block of the reference (r-index.hpp:134)
    std::vector<uint64_t> cum;          // cumulative ranks, sym_map order
    std::vector<std::pair<uint8_t, uint64_t>> runs; // (nuc code, length)
};

struct RiFile { // FastLocate as stored (src/r-index.cpp:266-376)
    uint32_t tag = 0, version = 0;
    uint64_t max_length = 0, flags = 0;
    IntVector samples, last_to_run;
    SdVector last, blocks_start_pos;
    uint8_t sym_map[256] = {0};
    std::vector<uint64_t> C;
    uint64_t sequence_size = 0;
    bool encoded = false, hasN = false;
    uint64_t enc_block_size = 10;
    std::vector<RefBlock> blocks;
    // COMPAT count_encoded / LF_encoded on an encoded index WITHOUT N: rankAt_encoded reads six cumulative
    // varints where five were written (src/r-index.cpp:578), so its run scan starts one varint late.  lit_runs[b] = the runs
    // the reference's scan sees in block b (code << 56 | length); empty unless the index has that shape.
    std::vector<std::vector<uint64_t>> lit_runs;
    uint64_t n_file_blocks = 0; // incl. a trailing never-filled block
    double ref_block_mean_bytes = 0;
    void parse(const uint8_t *p, uint64_t n);
};

struct TagFile { // TagArray as stored (src/tag_arrays.cpp:739-776)
    uint32_t format = 0;
    std::vector<uint64_t> items;     // raw item values in file order
    SdVector starts, bwt_intervals;
    void parse(const uint8_t *p, uint64_t n, uint32_t format_hint);
};

// what merge_tags takes from the graph (pgx_gbz.cpp)
struct GbzPaths {
    std::vector<uint64_t> first_node;        // per GBWT sequence: graph node id of its first node (0 = empty path)
    std::vector<uint32_t> component_of_node; // by node id; ~0u for ids that do not occur
    uint32_t n_components = 0;
    uint64_t max_node_id = 0;
};
void parse_gbz_paths(const std::string &path, GbzPaths &g);
void write_compact_tags(const char *out_path, const uint64_t *values, const uint64_t *lengths, uint64_t n_runs, uint64_t max_node_floor);

std::vector<uint8_t> read_whole_file(const std::string &path);
void write_whole_file(const std::string &path, const std::vector<uint8_t> &bytes);

// ------------------------------------------------------------------------------------------
struct HostImage {
    PgxConsts consts;
    std::vector<uint8_t> blocks;   // n_blocks * 64
    std::vector<uint64_t> dir;
    std::vector<uint64_t> bstart; // host only (tests): absolute block starts
    std::vector<uint16_t> blow;   // device: low dir_shift bits of each block start
    std::vector<uint32_t> exc;    // DENSE2: exception runs (pgx_image.h)
    std::vector<uint8_t> pairs;   // PAIRS image blocks (pgx_image.h), empty without one
    std::vector<uint64_t> sbase2, pbase; // WIDE images: superblock bases of the DENSE2 / PAIRS image (pgx_image.h)
    std::vector<uint64_t> tstart, tvals;
    std::vector<uint32_t> tdir;
    uint64_t n_runs = 0;
};

struct LitHostImage { // literal count image (quirk 3), built on first use
    std::vector<uint64_t> bstart, cum, runs; // block starts; six cumulative counts per block (nuc order); all runs
    std::vector<uint32_t> roff;              // first run of block b in runs (n_blocks + 1 entries)
    uint32_t code_of[256], cslot_of[256];    // symbol_to_code (r-index.hpp:664-668: unknown -> 0), sym_map
    uint64_t C[8];
    bool built = false;
};

struct LocHostImage { // locate image (pgx_image.h), built on first use
    PgxLocConsts consts;
    std::vector<uint64_t> rstart, rsamp, lpos, lnext;
    std::vector<uint32_t> rdir, ldir;
    bool built = false;
};

void build_rank_image(const RiFile &ri, uint32_t mode, HostImage &img);
void build_locate_image(const RiFile &ri, LocHostImage &loc);
void build_literal_image(const RiFile &ri, LitHostImage &lit);
void build_tag_image(const TagFile &tf, HostImage &img);
void build_ext_tables(const RiFile &ri, uint32_t mode, PgxConsts &c);

} // namespace pgx

// opaque handle
struct pgx_device_image;
struct pgx_index {
    pgx::RiFile ri;
    pgx::TagFile tags;
    bool has_tags = false, has_rank = false;
    uint32_t mode = 0;
    pgx::HostImage img;
    pgx::LocHostImage loc;
    pgx::LitHostImage lit;
    std::vector<pgx_device_image *> dev; // one per device ordinal (lazily filled)
};

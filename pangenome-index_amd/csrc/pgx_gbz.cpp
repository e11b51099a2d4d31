// pgx_gbz.cpp -- the part of a GBZ file that merge_tags needs (src/merge_tags.cpp:443-512): the GBWT's node records.
//
// The reference loads the whole GBZ (gbwtgraph::GBZ, simple-sds serialization) and asks it two things:
//   * gbz.index.extract(i)[0]                     the first node of path i            (merge_tags.cpp:508-515)
//   * gbwtgraph::weakly_connected_components       node id -> component               (algorithm.hpp:600-619)
// Both come out of the GBWT's compressed records alone (jltsiren/gbwt, not present under /root/reference; restated from its
// published file format, anchored on the reference's fixtures test_data/**/*.gbz: every path walked through the records
// ends at the endmarker after exactly header.size steps and the reverse paths mirror the forward ones, tests/test_gbz.py):
//   GBZ    = header {u32 tag "GBZ ", u32 version, u64 flags}, tags (StringArray), GBWT, GBWTGraph (not read)
//   GBWT   = header {u32 tag 0x6B376B37, u32 version, u64 sequences, size, offset, alphabet_size, flags}, tags (StringArray),
//            RecordArray {SparseVector of record starts, byte vector}, ... (document array samples, metadata: not read)
//   record = ByteCode outdegree; outdegree x (ByteCode node delta, ByteCode offset); runs of (edge rank, length) in
//            gbwt::Run coding: sigma >= 255: two ByteCodes (rank, length - 1); otherwise one byte = rank + sigma * (length - 1),
//            continued by a ByteCode when the basic length reaches 256 / sigma
//   simple-sds: every field padded to 8 bytes; vector = u64 count + items; bit vector = u64 ones, u64 bit length, word vector,
//            three optional supports (u64 size in words + body); int vector = u64 count, u64 width, u64 bit length, word vector;
//            sparse vector = u64 universe, high bit vector, low int vector; string array = sparse vector of starts, byte
//            vector alphabet, int vector of character codes
// Host only.
#include <algorithm>
#include <numeric>

#include "pgx_host.hpp"

using namespace pgx;

#define PGX_GUARD_BEGIN try {
#define PGX_GUARD_END                                                                               \
    }                                                                                               \
    catch (const pgx::Error &e) { pgx::set_last_error(e.what()); return e.code; }                   \
    catch (const std::bad_alloc &) { pgx::set_last_error("out of host memory"); return PGX_ERR_NOMEM; } \
    catch (const std::exception &e) { pgx::set_last_error(e.what()); return PGX_ERR_FORMAT; }

namespace {
struct Sds {
    const uint8_t *p;
    uint64_t n, o = 0;
    uint64_t u64(const char *what) {
        if (o + 8 > n) throw Error(PGX_ERR_FORMAT, std::string("GBZ: truncated file while reading ") + what);
        uint64_t v;
        std::memcpy(&v, p + o, 8);
        o += 8;
        return v;
    }
    void skip_words(uint64_t w, const char *what) {
        if (w > (n - o) / 8) throw Error(PGX_ERR_FORMAT, std::string("GBZ: truncated file while skipping ") + what);
        o += 8 * w;
    }
    std::vector<uint64_t> words(const char *what) {
        const uint64_t k = u64(what);
        if (k > (n - o) / 8) throw Error(PGX_ERR_FORMAT, std::string("GBZ: word vector longer than the file in ") + what);
        std::vector<uint64_t> v(k);
        if (k) std::memcpy(v.data(), p + o, 8 * k);
        o += 8 * k;
        return v;
    }
    // byte vector: returns the span, advances over the padding
    std::pair<const uint8_t *, uint64_t> bytes(const char *what) {
        const uint64_t k = u64(what);
        const uint64_t padded = (k + 7) / 8 * 8;
        if (padded > n - o) throw Error(PGX_ERR_FORMAT, std::string("GBZ: byte vector longer than the file in ") + what);
        const uint8_t *b = p + o;
        o += padded;
        return {b, k};
    }
    // sparse vector -> ascending positions of its ones
    std::vector<uint64_t> sparse(uint64_t &universe, const char *what) {
        universe = u64(what);
        const uint64_t ones = u64(what), hbits = u64(what);
        const std::vector<uint64_t> high = words(what);
        if (hbits > high.size() * 64) throw Error(PGX_ERR_FORMAT, std::string("GBZ: bad high part in ") + what);
        for (int i = 0; i < 3; i++) skip_words(u64(what), what); // rank / select / select_zero supports
        const uint64_t ln = u64(what), lw = u64(what), lbits = u64(what);
        const std::vector<uint64_t> low = words(what);
        if (ln != ones || lw > 64 || lbits != ln * lw || lbits > low.size() * 64) throw Error(PGX_ERR_FORMAT, std::string("GBZ: bad low part in ") + what);
        std::vector<uint64_t> out;
        out.reserve(ones);
        uint64_t zeros = 0;
        for (uint64_t pos = 0; pos < hbits; pos++) {
            if ((high[pos >> 6] >> (pos & 63)) & 1) {
                const uint64_t k = out.size();
                if (k >= ones) throw Error(PGX_ERR_FORMAT, std::string("GBZ: more ones than declared in ") + what);
                uint64_t lv = 0;
                if (lw) {
                    const uint64_t bit = k * lw, wd = bit >> 6, sh = bit & 63;
                    lv = low[wd] >> sh;
                    if (sh + lw > 64) lv |= low[wd + 1] << (64 - sh);
                    if (lw < 64) lv &= (1ull << lw) - 1;
                }
                out.push_back((zeros << lw) | lv);
            } else zeros++;
        }
        if (out.size() != ones) throw Error(PGX_ERR_FORMAT, std::string("GBZ: fewer ones than declared in ") + what);
        return out;
    }
    void skip_string_array(const char *what) {
        uint64_t uni;
        (void)sparse(uni, what);
        (void)bytes(what);                                  // alphabet
        (void)u64(what); (void)u64(what); (void)u64(what);  // int vector: count, width, bit length
        (void)words(what);
    }
};

inline uint64_t bc(const uint8_t *d, uint64_t end, uint64_t &o) { // gbwt::ByteCode
    uint64_t v = 0, sh = 0;
    for (;;) {
        if (o >= end || sh > 63) throw Error(PGX_ERR_FORMAT, "GBZ: bad ByteCode value in a GBWT record");
        const uint8_t b = d[o++];
        v |= (uint64_t)(b & 0x7F) << sh;
        if (!(b & 0x80)) return v;
        sh += 7;
    }
}
} // namespace

namespace pgx {
void parse_gbz_paths(const std::string &path, GbzPaths &g) {
    const std::vector<uint8_t> file = read_whole_file(path);
    Sds s{file.data(), file.size()};
    const uint64_t tagver = s.u64("GBZ header");
    if ((uint32_t)tagver != 0x205A4247u) throw Error(PGX_ERR_FORMAT, "GBZ: invalid tag (not a GBZ file)");
    (void)s.u64("GBZ flags");
    s.skip_string_array("GBZ tags");
    const uint64_t gtag = s.u64("GBWT header");
    if ((uint32_t)gtag != 0x6B376B37u) throw Error(PGX_ERR_FORMAT, "GBZ: GBWT tag not found where the simple-sds layout puts it");
    const uint64_t n_seq = s.u64("GBWT sequences");
    (void)s.u64("GBWT size");
    const uint64_t offset = s.u64("GBWT offset"), sigma = s.u64("GBWT alphabet size");
    (void)s.u64("GBWT flags");
    s.skip_string_array("GBWT tags");
    uint64_t universe = 0;
    const std::vector<uint64_t> starts = s.sparse(universe, "GBWT record index");
    const auto data = s.bytes("GBWT records");
    if (universe != data.second || sigma < offset || starts.size() != sigma - offset || starts.empty())
        throw Error(PGX_ERR_FORMAT, "GBZ: GBWT record index does not match its header");
    const uint8_t *d = data.first;
    const uint64_t n_rec = starts.size();
    const uint64_t max_node_id = (sigma - 1) / 2; // GBWT node = 2 * id + orientation
    // union-find over graph node ids; edges = the outgoing edges of every record (what GBWTGraph::follow_edges follows)
    std::vector<uint64_t> parent(max_node_id + 1);
    std::iota(parent.begin(), parent.end(), 0);
    std::vector<uint8_t> present(max_node_id + 1, 0);
    auto find = [&](uint64_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    g.first_node.assign(n_seq, 0);
    for (uint64_t r = 0; r < n_rec; r++) {
        uint64_t o = starts[r];
        const uint64_t end = r + 1 < n_rec ? starts[r + 1] : data.second;
        if (o > end || end > data.second) throw Error(PGX_ERR_FORMAT, "GBZ: GBWT record offsets not monotone");
        if (o == end) continue; // node without a record
        const uint64_t node = r == 0 ? 0 : r + offset; // record 0 is the endmarker
        const uint64_t outdeg = bc(d, end, o);
        std::vector<uint64_t> succ(outdeg);
        uint64_t prev = 0;
        for (uint64_t e = 0; e < outdeg; e++) {
            prev += bc(d, end, o);
            (void)bc(d, end, o); // offset in the successor's record
            succ[e] = prev;
            if (prev > 2 * max_node_id + 1) throw Error(PGX_ERR_FORMAT, "GBZ: edge to a node beyond the alphabet");
        }
        if (node) {
            if (outdeg == 0 && o >= end) continue; // a record without edges or visits: no path uses the node, GBWTGraph does not have it
            present[node / 2] = 1;
            for (uint64_t t : succ)
                if (t) { present[t / 2] = 1; const uint64_t a = find(node / 2), b = find(t / 2); if (a != b) parent[std::max(a, b)] = std::min(a, b); }
            continue;
        }
        // endmarker record: position i of its body = the first node of sequence i
        const uint64_t rc = (outdeg && outdeg < 255) ? 256 / outdeg : 0;
        uint64_t seq = 0;
        while (o < end && seq < n_seq) {
            uint64_t rank, len;
            if (rc == 0) { rank = bc(d, end, o); len = bc(d, end, o) + 1; }
            else {
                const uint8_t code = d[o++];
                rank = code % outdeg; len = code / outdeg + 1;
                if (len >= rc) len += bc(d, end, o);
            }
            if (rank >= outdeg) throw Error(PGX_ERR_FORMAT, "GBZ: run of an edge the endmarker does not have");
            for (uint64_t k = 0; k < len && seq < n_seq; k++) g.first_node[seq++] = succ[rank] / 2; // 0 for an empty path
        }
        if (seq != n_seq) throw Error(PGX_ERR_FORMAT, "GBZ: the endmarker record is shorter than the number of sequences");
    }
    // components numbered by their smallest node id (gbwtgraph::weakly_connected_components order)
    g.component_of_node.assign(max_node_id + 1, ~0u);
    uint32_t n_comp = 0;
    for (uint64_t v = 1; v <= max_node_id; v++) {
        if (!present[v]) continue;
        const uint64_t root = find(v);
        if (g.component_of_node[root] == ~0u) g.component_of_node[root] = n_comp++; // root = smallest id of its component: met first
        g.component_of_node[v] = g.component_of_node[root];
    }
    g.n_components = n_comp;
    g.max_node_id = 0;
    for (uint64_t v = max_node_id; v >= 1; v--) if (present[v]) { g.max_node_id = v; break; }
}
} // namespace pgx

extern "C" pgx_status pgx_gbz_paths(const char *gbz_path, uint64_t *n_sequences, uint64_t *first_node, uint32_t *component, uint64_t cap,
                                    uint64_t *max_node_id, uint32_t *n_components) {
    PGX_GUARD_BEGIN
    if (!gbz_path || !n_sequences) throw Error(PGX_ERR_ARG, "pgx_gbz_paths: null argument");
    GbzPaths g;
    parse_gbz_paths(gbz_path, g);
    *n_sequences = g.first_node.size();
    if (max_node_id) *max_node_id = g.max_node_id;
    if (n_components) *n_components = g.n_components;
    for (uint64_t i = 0; i < g.first_node.size() && i < cap; i++) {
        if (first_node) first_node[i] = g.first_node[i];
        if (component) component[i] = g.first_node[i] ? g.component_of_node[g.first_node[i]] : ~0u;
    }
    return PGX_OK;
    PGX_GUARD_END
}

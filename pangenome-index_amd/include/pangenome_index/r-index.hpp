// pangenome_index/r-index.hpp -- drop-in host mirror of the query side of panindexer::FastLocate
// (reference include/pangenome_index/r-index.hpp:44-716) implemented over the C ABI of libpgx.
//
// Same names, argument meaning and error behaviour as the reference for the find_mems path:
//   load_encoded / load            throw std::runtime_error("FastLocate: Invalid tag" ...) like
//                                  sdsl::simple_sds::InvalidData (src/r-index.cpp:412-419)
//   backward/forward_extend[_encoded], bi_interval, complement, sym_map, C, bwt_size, tot_strings
// Every query runs on the GPU; a per-call extension costs a kernel launch, so batch callers should
// use panindexer::find_all_mems_batch (algorithm.hpp) or pgx_extend_batch directly.
#ifndef PANGENOME_INDEX_R_INDEX_HPP
#define PANGENOME_INDEX_R_INDEX_HPP

#include <array>
#include <cstdint>
#include <istream>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/pgx.h"
#include "utils.hpp"

namespace panindexer {

typedef std::uint64_t size_type;

class FastLocate {
public:
    // r-index.hpp:118-130
    struct bi_interval {
        size_t forward; // k
        size_t reverse; // l
        int64_t size;   // s
        bi_interval() = default;
        bi_interval(size_type fwd, size_type rev, size_type sz) : forward(fwd), reverse(rev), size((int64_t)sz) {}
        bool operator==(const bi_interval &o) const { return forward == o.forward && reverse == o.reverse && size == o.size; }
    };

    std::array<uint8_t, 256> sym_map{}; // r-index.hpp:307
    std::vector<size_t> C;              // r-index.hpp:310

    FastLocate() { initialize_complement_table(); }
    FastLocate(const FastLocate &) = delete;
    FastLocate &operator=(const FastLocate &) = delete;
    ~FastLocate() { if (h_) pgx_index_close(h_); }

    // src/r-index.cpp:406-459: falls back to the legacy layout when the ENCODED_BLOCKS flag is clear
    void load_encoded(std::istream &in) { open(in); }
    void load(std::istream &in) { open(in); } // src/r-index.cpp:378-404

    bool is_encoded() const { return info_.is_encoded != 0; }            // r-index.hpp:409
    size_t bwt_size() const { return info_.bwt_size; }                   // r-index.hpp:568
    size_t get_sequence_size() const { return info_.bwt_size; }
    size_t tot_strings() const { return C.size() > 1 ? C[1] - C[0] : 0; } // r-index.hpp:484
    size_t text_size() const { return bwt_size() - tot_strings(); }

    void initialize_complement_table() { // src/r-index.cpp:1512-1529
        for (size_t i = 0; i < 256; ++i) complement_table[i] = (uint8_t)i;
        complement_table['A'] = 'T'; complement_table['C'] = 'G'; complement_table['G'] = 'C'; complement_table['T'] = 'A';
        complement_table['a'] = 't'; complement_table['c'] = 'g'; complement_table['g'] = 'c'; complement_table['t'] = 'a';
    }
    inline size_t complement(size_t symbol) const { return complement_table[symbol & 0xFF]; }

    // src/r-index.cpp:713-764 (legacy twins :1395-1428, :1500-1509: same entry points here, the
    // layout difference is carried by the device image)
    bi_interval backward_extend_encoded(const bi_interval &b, size_t symbol) { return extend(b, symbol, false); }
    bi_interval forward_extend_encoded(const bi_interval &b, size_t symbol) { return extend(b, symbol, true); }
    bi_interval backward_extend(const bi_interval &b, size_t symbol) { return extend(b, symbol, false); }
    bi_interval forward_extend(const bi_interval &b, size_t symbol) { return extend(b, symbol, true); }

    // r-index.hpp:540-556: unidirectional backward search; the empty range is {1, 0}
    std::pair<size_t, size_t> count_encoded(std::string &pattern) const { return count_impl(pattern); }
    std::pair<size_t, size_t> count(std::string &pattern) { return count_impl(pattern); }

    // src/r-index.cpp:650-711: one LF step of an inclusive BWT range ({1, 0} = empty).  The reference's out-parameters
    // starts_with_to / first_run are set to false / max like its own implementation does (:658-659, :693-694).
    std::pair<size_type, size_type> LF(std::pair<size_type, size_type> range, size_t sym) const { return lf_impl(range, sym); }
    std::pair<size_type, size_type> LF_encoded(std::pair<size_type, size_type> range, size_t sym) const { return lf_impl(range, sym); }
    std::pair<size_type, size_type> LF(std::pair<size_type, size_type> range, size_t sym, bool &starts_with_to, size_t &first_run) const {
        starts_with_to = false; first_run = ~(size_t)0;
        return lf_impl(range, sym);
    }
    std::pair<size_type, size_type> LF_encoded(std::pair<size_type, size_type> range, size_t sym, bool &starts_with_to, size_t &first_run) const {
        starts_with_to = false; first_run = ~(size_t)0;
        return lf_impl(range, sym);
    }

    // ---- locate (r-index.hpp:385-406, 424-436, 490-501, 616-618; src/r-index.cpp:1252-1366) ----
    typedef std::pair<size_type, size_type> range_type; // gbwt::range_type: inclusive BWT range, empty when second < first
    constexpr static size_type NO_POSITION = ~(size_type)0;
    size_type pack(size_type seq_id, size_type seq_offset) const { return seq_id * info_.max_length + seq_offset; }
    size_type seqId(size_type offset) const { return offset / info_.max_length; }
    size_type seqOffset(size_type offset) const { return offset % info_.max_length; }
    std::pair<size_type, size_type> unpack(size_type offset) const { return std::make_pair(seqId(offset), seqOffset(offset)); }
    size_type size() const { return info_.n_samples; } // runs
    size_type getSample(size_type run_id) const {
        const void *ptr = nullptr;
        uint64_t bytes = 0;
        if (!h_ || pgx_index_image_view(h_, 9, &ptr, &bytes) != PGX_OK) throw std::runtime_error(pgx_last_error());
        if (run_id >= bytes / 8) throw std::out_of_range("FastLocate::getSample");
        return static_cast<const uint64_t *>(ptr)[run_id];
    }
    size_type locateFirst() const { return getSample(0); }
    size_type locateNext(size_type prev) const {
        uint64_t out = NO_POSITION;
        if (!h_ || pgx_locate_next_batch(h_, device_, &prev, 1, &out) != PGX_OK) throw std::runtime_error(pgx_last_error());
        return out;
    }
    size_type locate_next_nth(size_type prev, size_type n) const {
        for (size_type i = 0; i < n; i++) prev = locateNext(prev);
        return prev;
    }
    // sorted unique sequence ids of the suffixes in `state` (the `first` hint of the reference is not needed here)
    std::vector<size_type> locate(range_type state, size_type = NO_POSITION) const { return locate_impl(state); }
    std::vector<size_type> locate_encoded(range_type state, size_type = NO_POSITION) const { return locate_impl(state); }
    // many ranges in one launch: offsets has ranges.size() + 1 entries
    void locate_batch(const std::vector<range_type> &states, std::vector<size_type> &offsets, std::vector<size_type> &seq_ids) const {
        std::vector<uint64_t> f(states.size()), l(states.size());
        uint64_t cap = 0;
        for (size_t i = 0; i < states.size(); i++) {
            f[i] = states[i].first; l[i] = states[i].second;
            if (l[i] >= f[i]) cap += l[i] - f[i] + 1;
        }
        offsets.assign(states.size() + 1, 0);
        seq_ids.assign(cap ? cap : 1, 0);
        if (!h_ || pgx_locate_batch(h_, device_, f.data(), l.data(), states.size(), PGX_LOCATE_SEQ_IDS | PGX_LOCATE_UNIQUE, offsets.data(),
                                    seq_ids.data(), cap) != PGX_OK)
            throw std::runtime_error(pgx_last_error());
        seq_ids.resize(offsets.back());
    }
    std::vector<size_type> decompressSA() const { return decompress(0); }                   // src/r-index.cpp:1343-1353
    std::vector<size_type> decompressDA() const { return decompress(PGX_LOCATE_SEQ_IDS); }  // :1355-1361
    std::vector<size_type> decompressSA_encoded() const { return decompressSA(); }          // r-index.hpp:397
    std::vector<size_type> decompressDA_encoded() const { return decompressDA(); }          // r-index.hpp:398

    // library handle (for the batch entry points of algorithm.hpp)
    pgx_index *handle() const { return h_; }
    int device() const { return device_; }
    void set_device(int d) { device_ = d; }
    void set_mode(uint32_t m) { mode_ = m; } // PGX_MODE_COMPAT (default) / PGX_MODE_STRICT; before load

private:
    pgx_index *h_ = nullptr;
    pgx_index_info info_{};
    int device_ = 0;
    uint32_t mode_ = PGX_MODE_COMPAT;
    std::array<uint8_t, 256> complement_table{};

    void open(std::istream &in) {
        std::vector<char> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (h_) { pgx_index_close(h_); h_ = nullptr; }
        if (pgx_index_open_memory(bytes.data(), bytes.size(), nullptr, 0, PGX_TAGS_AUTO, mode_, &h_) != PGX_OK)
            throw std::runtime_error(pgx_last_error());
        pgx_index_info_get(h_, &info_);
        uint64_t c[8];
        pgx_index_tables(h_, sym_map.data(), c, nullptr);
        C.assign(c, c + info_.sigma);
    }

    std::vector<size_type> locate_impl(range_type state) const {
        std::vector<size_type> offsets, ids;
        locate_batch(std::vector<range_type>(1, state), offsets, ids);
        return ids;
    }

    std::vector<size_type> decompress(uint32_t flags) const {
        if (!h_) throw std::runtime_error("FastLocate: no index loaded");
        std::vector<size_type> out(info_.bwt_size);
        if (pgx_decompress_sa(h_, device_, flags, out.data()) != PGX_OK) throw std::runtime_error(pgx_last_error());
        return out;
    }

    std::pair<size_type, size_type> lf_impl(std::pair<size_type, size_type> range, size_t sym) const {
        if (!h_) throw std::runtime_error("FastLocate: no index loaded");
        const pgx_range in{range.first, range.second};
        pgx_range out{1, 0};
        const uint8_t s = (uint8_t)sym;
        if (pgx_lf_batch(h_, device_, &in, &s, 1, &out) != PGX_OK) throw std::runtime_error(pgx_last_error());
        return {out.first, out.second};
    }

    std::pair<size_t, size_t> count_impl(const std::string &pattern) const {
        if (!h_) throw std::runtime_error("FastLocate: no index loaded");
        const uint64_t offs[2] = {0, pattern.size()};
        pgx_range r{1, 0};
        if (pgx_count_batch(h_, device_, reinterpret_cast<const uint8_t *>(pattern.data()), offs, 1, &r) != PGX_OK)
            throw std::runtime_error(pgx_last_error());
        return {(size_t)r.first, (size_t)r.second};
    }

    bi_interval extend(const bi_interval &b, size_t symbol, bool fwd) {
        if (!h_) throw std::runtime_error("FastLocate: no index loaded");
        pgx_biint in{b.forward, b.reverse, b.size}, out{0, 0, 0};
        const uint8_t sym = (uint8_t)symbol, f = fwd ? 1 : 0;
        if (pgx_extend_batch(h_, device_, &in, &sym, &f, 1, &out) != PGX_OK) throw std::runtime_error(pgx_last_error());
        return bi_interval(out.forward, out.reverse, (size_type)out.size);
    }
};

} // namespace panindexer

#endif // PANGENOME_INDEX_R_INDEX_HPP

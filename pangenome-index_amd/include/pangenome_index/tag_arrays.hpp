// pangenome_index/tag_arrays.hpp -- drop-in host mirror of the query side of panindexer::TagArray
// (reference include/pangenome_index/tag_arrays.hpp:27-128) over the C ABI of libpgx.
//
//   load_compressed_tags            ByteCode format   (src/tag_arrays.cpp:739-763)
//   load_compressed_tags_sdsl/_compact  sdsl compact  (src/tag_arrays.cpp:766-776, alias hpp:68)
//   query_compressed / query_compressed_compact (src/tag_arrays.cpp:780-890): number_of_runs by
//   reference, positions printed to std::cout in the reference's grammar
//   ("Number of unique positions: N\n" then "p, p, ..., \n")
#ifndef PANGENOME_INDEX_TAG_ARRAYS_HPP
#define PANGENOME_INDEX_TAG_ARRAYS_HPP

#include <cstdint>
#include <iostream>
#include <istream>
#include <iterator>
#include <stdexcept>
#include <tuple>
#include <utility>
#include <vector>

#include "../../../include/pgx.h"

namespace panindexer {

typedef std::tuple<int64_t, bool, size_t> pos_t; // libhandlegraph pos_t: (node id, is_rev, offset)
inline pos_t make_pos_t(int64_t id, bool is_rev, size_t off) { return std::make_tuple(id, is_rev, off); }

class TagArray {
public:
    TagArray() = default;
    TagArray(const TagArray &) = delete;
    TagArray &operator=(const TagArray &) = delete;
    ~TagArray() { if (h_) pgx_index_close(h_); }

    void load_compressed_tags(std::istream &in) { open(in, PGX_TAGS_BYTECODE); }
    void load_compressed_tags_sdsl(std::istream &in) { open(in, PGX_TAGS_COMPACT); }
    void load_compressed_tags_compact(std::istream &in) { open(in, PGX_TAGS_COMPACT); } // tag_arrays.hpp:68
    void load_any(std::istream &in) { open(in, PGX_TAGS_AUTO); }                        // format sniffing (ours)

    void query_compressed(size_t start, size_t end, size_t &number_of_runs) { query(start, end, number_of_runs, true); }
    void query_compressed_compact(size_t start, size_t end, size_t &number_of_runs) { query(start, end, number_of_runs, true); }
    // same without the printing
    std::vector<uint64_t> query_positions(size_t start, size_t end, size_t &number_of_runs) {
        return query(start, end, number_of_runs, false);
    }

    // src/tag_arrays.cpp:59-70 (length_bits = 9, tag_arrays.hpp:116) and :47-55
    static std::pair<pos_t, uint16_t> decode_run(uint64_t encoded) {
        return std::make_pair(make_pos_t((int64_t)(encoded >> 20), (encoded >> 10) & 1, encoded & 0x3FF), (uint16_t)((encoded >> 11) & 0x1FF));
    }
    static pos_t decode_run_length_compact(uint64_t encoded) { return make_pos_t((int64_t)(encoded >> 11), (encoded >> 10) & 1, encoded & 0x3FF); }

    pgx_index *handle() const { return h_; }
    void set_device(int d) { device_ = d; }

private:
    pgx_index *h_ = nullptr;
    int device_ = 0;

    void open(std::istream &in, uint32_t fmt) {
        std::vector<char> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (h_) { pgx_index_close(h_); h_ = nullptr; }
        if (pgx_index_open_memory(nullptr, 0, bytes.data(), bytes.size(), fmt, PGX_MODE_COMPAT, &h_) != PGX_OK)
            throw std::runtime_error(pgx_last_error());
    }

    std::vector<uint64_t> query(size_t start, size_t end, size_t &number_of_runs, bool print) {
        if (!h_) throw std::runtime_error("TagArray: no tags loaded");
        uint64_t st = start, en = end, rn = 0, po[2] = {0, 0}, over = 0;
        if (pgx_tag_query_batch(h_, device_, &st, &en, 1, &rn, po, nullptr, 0, &over) != PGX_OK) throw std::runtime_error(pgx_last_error());
        std::vector<uint64_t> pos(po[1] ? po[1] : 1);
        if (pgx_tag_query_batch(h_, device_, &st, &en, 1, &rn, po, pos.data(), pos.size(), &over) != PGX_OK)
            throw std::runtime_error(pgx_last_error());
        pos.resize(po[1]);
        number_of_runs = rn;
        if (print) { // src/tag_arrays.cpp:885-889
            std::cout << "Number of unique positions: " << pos.size() << std::endl;
            for (auto p : pos) std::cout << p << ", ";
            std::cout << std::endl;
        }
        return pos;
    }
};

} // namespace panindexer

#endif // PANGENOME_INDEX_TAG_ARRAYS_HPP

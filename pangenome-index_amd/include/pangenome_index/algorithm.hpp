// pangenome_index/algorithm.hpp -- drop-in host mirror of the query side of the reference's
// include/pangenome_index/algorithm.hpp:644-757 (struct MEM, find_all_mems) over libpgx, plus the
// batch overloads a GPU needs (one call per read is pointless on a device).
#ifndef PANGENOME_INDEX_ALGORITHM_HPP
#define PANGENOME_INDEX_ALGORITHM_HPP

#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "r-index.hpp"
#include "tag_arrays.hpp"

namespace panindexer {

// algorithm.hpp:644-651
struct MEM {
    size_t start;
    size_t end;
    size_t bwt_start;
    int64_t size;
};

// algorithm.hpp:653-736: one call of the MEM finder at start position x; pushes at most one MEM, returns the next start.
// (A device launch per call: for real work use find_all_mems_batch, which runs the whole loop of :745-748 on the device.)
inline size_t find_mems_function(const std::string &pattern, size_t min_len, size_t min_occ, size_t x, FastLocate &fmd_index,
                                 std::vector<MEM> &output) {
    const uint64_t offs[2] = {0, pattern.size()}, read_of = 0, xs = x;
    uint64_t next_x = 0;
    pgx_mem m{};
    uint8_t has = 0;
    if (pgx_find_mems_function_batch(fmd_index.handle(), fmd_index.device(), reinterpret_cast<const uint8_t *>(pattern.data()), offs, 1, &read_of,
                                     &xs, 1, min_len, min_occ, &next_x, &m, &has, nullptr) != PGX_OK)
        throw std::runtime_error(pgx_last_error());
    if (has) output.push_back(MEM{(size_t)m.start, (size_t)m.end, (size_t)m.bwt_start, m.size});
    return (size_t)next_x;
}

// find_all_mems for many reads in one device batch; result[i] = MEMs of reads[i] in discovery order
inline std::vector<std::vector<MEM>> find_all_mems_batch(const std::vector<std::string> &reads, size_t min_len, size_t min_occ,
                                                         FastLocate &fmd_index) {
    std::string cat;
    std::vector<uint64_t> offs(reads.size() + 1, 0);
    for (size_t i = 0; i < reads.size(); i++) { cat += reads[i]; offs[i + 1] = cat.size(); }
    pgx_batch *b = nullptr;
    pgx_result r{};
    if (pgx_find_mems_batch(fmd_index.handle(), fmd_index.device(), reinterpret_cast<const uint8_t *>(cat.data()), offs.data(), reads.size(),
                            min_len, min_occ, 0, &b, &r) != PGX_OK)
        throw std::runtime_error(pgx_last_error());
    std::vector<std::vector<MEM>> out(reads.size());
    for (size_t i = 0; i < reads.size(); i++)
        for (uint64_t m = r.mem_offsets[i]; m < r.mem_offsets[i + 1]; m++)
            out[i].push_back(MEM{(size_t)r.mems[m].start, (size_t)r.mems[m].end, (size_t)r.mems[m].bwt_start, r.mems[m].size});
    pgx_batch_free(b);
    return out;
}

// algorithm.hpp:739-757 (logs one line per read to stderr like the reference, :754)
inline std::vector<MEM> find_all_mems(const std::string &pattern, size_t min_len, size_t min_occ, FastLocate &fmd_index) {
    std::vector<MEM> mems = find_all_mems_batch(std::vector<std::string>{pattern}, min_len, min_occ, fmd_index)[0];
    std::cerr << "[find_all_mems] total mems=" << mems.size() << std::endl;
    return mems;
}

} // namespace panindexer

#endif // PANGENOME_INDEX_ALGORITHM_HPP

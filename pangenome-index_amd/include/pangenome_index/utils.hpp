// pangenome_index/utils.hpp -- constants of the reference (include/pangenome_index/utils.hpp:9-11)
#ifndef PANGENOME_INDEX_UTILS_HPP
#define PANGENOME_INDEX_UTILS_HPP

#include <vector>

namespace panindexer {
constexpr char NENDMARKER = '\n';
const std::vector<char> nuc = {NENDMARKER, 'A', 'C', 'G', 'N', 'T'}; // code order of the 3-bit run codes
} // namespace panindexer

#endif

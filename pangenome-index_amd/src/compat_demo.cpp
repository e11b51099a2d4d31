// compat_demo -- the reference's find_mems main() written against the drop-in C++ headers
// (FastLocate / TagArray / find_all_mems), one read at a time exactly like src/find_mems.cpp:94-139.
// Its stdout must equal the batched find_mems CLI (tests/test_gpu_cli.py).
#include <fstream>
#include <iostream>

#include "../include/pangenome_index/algorithm.hpp"
#include "../include/pangenome_index/tag_arrays.hpp"

using namespace panindexer;

int main(int argc, char **argv) {
    if (argc < 6) return 1;
    FastLocate r_index;
    {
        std::ifstream rin(argv[1], std::ios::binary);
        if (!rin) { std::cerr << "Cannot open r-index: " << argv[1] << std::endl; std::exit(EXIT_FAILURE); }
        r_index.load_encoded(rin);
    }
    TagArray tag_array;
    std::ifstream in_ds(argv[2], std::ios::binary);
    tag_array.load_any(in_ds);
    std::ifstream reads(argv[3]);
    size_t mem_length = std::stoi(argv[4]), min_occ = std::stoi(argv[5]);
    std::string read;
    int i = 0;
    // sanity of the public members the reference exposes
    FastLocate::bi_interval full(0, 0, r_index.bwt_size());
    auto a = r_index.backward_extend(full, 'A');
    std::cerr << "sigma=" << r_index.C.size() << " sym_map[A]=" << (int)r_index.sym_map['A'] << " bwd(A)=" << a.forward << "," << a.reverse
              << "," << a.size << " comp(A)=" << (char)r_index.complement('A') << " strings=" << r_index.tot_strings() << std::endl;
    // locate side of the same class (r-index.hpp:385-406,490-501): Locate_* of tests/test_rindex.cpp use exactly these calls
    {
        const auto first = r_index.locateFirst();
        const auto da = r_index.decompressDA_encoded();
        const auto ids = r_index.locate(FastLocate::range_type(0, r_index.tot_strings() - 1));
        std::cerr << "locate: first=" << first << " next=" << r_index.locateNext(first) << " seq(first)=" << r_index.seqId(first) << "+"
                  << r_index.seqOffset(first) << " DA[0..3]=" << da[0] << "," << da[1] << "," << da[2] << "," << da[3] << " |DA|=" << da.size()
                  << " locate(endmarkers)=" << ids.size() << ":" << ids.front() << ".." << ids.back() << std::endl;
    }
    while (std::getline(reads, read)) {
        if (read.empty()) continue;
        i++;
        auto mems = find_all_mems(read, mem_length, min_occ, r_index);
        std::cout << "Seq: " << i << std::endl;
        for (const auto &mem : mems) {
            std::cout << "MEM START: " << mem.start << ", MEM END: " << mem.end << " BWT START: " << mem.bwt_start << " SIZE: " << mem.size << std::endl;
            size_t tag_nums = 0;
            tag_array.query_compressed_compact(mem.bwt_start, mem.bwt_start + mem.size - 1, tag_nums);
        }
        std::cout << std::endl;
    }
    return 0;
}

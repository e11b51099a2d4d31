// convert_tags -- the reference CLI (src/convert_tags.cpp:13-40): build_tags' "algorithm format" -> a query format.
//
//   convert_tags <input_algorithm_file> <output_compressed_file> [encoded_starts_tmp] [bwt_intervals_tmp] [--format compact|bytecode]
//
// compact (default) is the sdsl-compact format find_mems loads (src/find_mems.cpp:79); bytecode is the ByteCode query format
// of TagArray::load_compressed_tags (the reference's xy_bidirectional_compressed.tags fixture, reproduced byte for byte).
// The two temporary-file arguments of the reference are accepted and unused.  Host only: no GPU is needed.
#include <cstdlib>
#include <iostream>
#include <string>

#include "../../include/pgx.h"

int main(int argc, char **argv) {
    if (argc < 3) {
        std::cerr << "Usage: " << argv[0] << " <input_algorithm_file> <output_compressed_file> [encoded_starts_tmp] [bwt_intervals_tmp]"
                  << " [--format compact|bytecode]" << std::endl;
        return EXIT_FAILURE;
    }
    int compact = 1;
    for (int i = 3; i < argc; i++)
        if (std::string(argv[i]) == "--format" && i + 1 < argc) compact = std::string(argv[++i]) == "bytecode" ? 0 : 1;
    if (pgx_convert_tags(argv[1], argv[2], compact) != PGX_OK) {
        std::cerr << pgx_last_error() << std::endl;
        return EXIT_FAILURE;
    }
    return 0;
}

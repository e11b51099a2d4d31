// build_rindex -- the reference CLI (src/build_rindex.cpp): run-length BWT (grlBWT .rl_bwt) -> encoded .ri on stdout.
//
//   build_rindex <file.rl_bwt> [--legacy] > out.ri
//
// The file is byte-identical to what the reference writes for the same input (tests/test_formats.py reproduces both of the
// reference's own .ri fixtures).  --legacy writes FastLocate::serialize's layout instead of serialize_encoded's.
// Host only: no GPU is needed.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "../../include/pgx.h"

int main(int argc, char **argv) {
    if (argc < 2) {
        std::cerr << "usage: build_rindex <file.rl_bwt> [--legacy] > out.ri" << std::endl;
        return EXIT_FAILURE;
    }
    const bool legacy = argc > 2 && std::string(argv[2]) == "--legacy";
    const std::string tmp = std::string(argv[1]) + ".ri.tmp";
    if (pgx_build_rindex(argv[1], tmp.c_str(), legacy ? 0 : 1) != PGX_OK) {
        std::cerr << pgx_last_error() << std::endl;
        return EXIT_FAILURE;
    }
    std::ifstream in(tmp, std::ios::binary);
    std::cout << in.rdbuf();
    std::cout.flush();
    in.close();
    std::remove(tmp.c_str());
    return 0;
}

// find_mems -- drop-in for the reference CLI (src/find_mems.cpp:13-147) on MI355X.
//
//   find_mems <r_index.ri> <tags> <reads.txt> <min_mem_length> <min_occ> [options after the 5 positionals]
//
// stdout grammar is the reference's, byte for byte (find_mems.cpp:115-118,138,144-145 and
// tag_arrays.cpp:885-889); reads are processed in device batches but printed in file order.
// Options (ours): --device N, --mode compat|strict, --batch N (reads per device batch),
//                 --tags-format auto|bytecode|compact, --quiet (no per-read stderr line)
// The tag file may be either query format; the reference's find_mems only loads the sdsl-compact one.
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/pgx.h"

int main(int argc, char **argv) {
    if (argc < 6) {
        std::cerr << "usage: find_mems <r_index.ri> <tags> <reads.txt> <min_mem_length> <min_occ>"
                     " [--device N] [--mode compat|strict] [--batch N] [--tags-format auto|bytecode|compact] [--quiet]" << std::endl;
        return EXIT_FAILURE;
    }
    const std::string r_index_file = argv[1], tag_array_index = argv[2], reads_file = argv[3];
    const size_t mem_length = (size_t)std::stoi(argv[4]); // find_mems.cpp:17 (std::stoi -> size_t)
    const size_t min_occ = (size_t)std::stoi(argv[5]);
    int device = 0;
    uint32_t mode = PGX_MODE_COMPAT, tfmt = PGX_TAGS_AUTO;
    size_t batch_reads = 1u << 20;
    bool quiet = false;
    for (int i = 6; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; std::exit(EXIT_FAILURE); } return argv[++i]; };
        if (a == "--device") device = std::stoi(next());
        else if (a == "--mode") mode = next() == "strict" ? PGX_MODE_STRICT : PGX_MODE_COMPAT;
        else if (a == "--batch") batch_reads = (size_t)std::stoull(next());
        else if (a == "--tags-format") { const std::string f = next(); tfmt = f == "bytecode" ? PGX_TAGS_BYTECODE : f == "compact" ? PGX_TAGS_COMPACT : PGX_TAGS_AUTO; }
        else if (a == "--quiet") quiet = true;
        else { std::cerr << "unknown option " << a << std::endl; return EXIT_FAILURE; }
    }
    if (batch_reads == 0) batch_reads = 1;

    auto time1 = std::chrono::high_resolution_clock::now();
    double total_mem_time = 0.0, total_tag_time = 0.0;

    std::cerr << "Reading the rindex file (encoded)" << std::endl;
    {
        std::ifstream rin(r_index_file, std::ios::binary);
        if (!rin) { std::cerr << "Cannot open r-index: " << r_index_file << std::endl; std::exit(EXIT_FAILURE); } // :30
    }
    {
        std::ifstream tin(tag_array_index, std::ios::binary);
        if (!tin) { std::cerr << "Cannot open tag array: " << tag_array_index << std::endl; std::exit(EXIT_FAILURE); }
    }
    pgx_index *h = nullptr;
    if (pgx_index_open(r_index_file.c_str(), tag_array_index.c_str(), tfmt, mode, &h) != PGX_OK) {
        std::cerr << pgx_last_error() << std::endl; // reference: uncaught sdsl::simple_sds::InvalidData
        return EXIT_FAILURE;
    }
    auto time2 = std::chrono::high_resolution_clock::now();
    std::cerr << "Loading r-index into memory took " << std::chrono::duration<double>(time2 - time1).count() << " seconds" << std::endl;
    std::cerr << "Reading the tag array index" << std::endl;
    if (pgx_index_to_device(h, device) != PGX_OK) { std::cerr << pgx_last_error() << std::endl; return EXIT_FAILURE; }
    auto time3 = std::chrono::high_resolution_clock::now();
    std::cerr << "Loading tag arrays took " << std::chrono::duration<double>(time3 - time2).count() << " seconds" << std::endl;

    std::ifstream reads(reads_file);
    if (!reads) { std::cerr << "Cannot open reads file: " << reads_file << std::endl; std::exit(EXIT_FAILURE); } // :91

    pgx_batch *b = nullptr;
    std::string cat, line, out;
    std::vector<uint64_t> offs;
    size_t seq_no = 0;
    bool eof = false;
    while (!eof) {
        cat.clear();
        offs.assign(1, 0);
        while (offs.size() <= batch_reads) {
            if (!std::getline(reads, line)) { eof = true; break; }
            if (line.empty()) continue; // :97
            cat += line;
            offs.push_back(cat.size());
        }
        const size_t n = offs.size() - 1;
        if (n == 0) break;
        // one long-lived batch: its device and pinned host buffers are reused by every chunk of reads
        pgx_result r;
        const uint8_t *rp = reinterpret_cast<const uint8_t *>(cat.data());
        pgx_status st = b ? pgx_batch_upload(b, rp, offs.data(), n) : pgx_batch_create(h, device, rp, offs.data(), n, &b);
        if (st == PGX_OK) st = pgx_batch_run(b, mem_length, min_occ, PGX_RUN_TAGS | PGX_RUN_TIMING, nullptr);
        if (st == PGX_OK) st = pgx_batch_result(b, &r);
        if (st != PGX_OK) {
            std::cerr << pgx_last_error() << std::endl;
            return EXIT_FAILURE;
        }
        pgx_timing t;
        if (pgx_batch_timing(b, &t) == PGX_OK) {
            total_mem_time += 1e-3 * (t.ms_find_mems + t.ms_compact);
            total_tag_time += 1e-3 * (t.ms_tag_locate + t.ms_tag_gather + t.ms_tag_sort);
        }
        out.clear();
        std::string err;
        for (size_t i = 0; i < n; i++) {
            ++seq_no;
            if (!quiet) err += "[find_all_mems] total mems=" + std::to_string(r.mem_offsets[i + 1] - r.mem_offsets[i]) + "\n"; // algorithm.hpp:754
            out += "Seq: " + std::to_string(seq_no) + "\n"; // :115
            for (uint64_t m = r.mem_offsets[i]; m < r.mem_offsets[i + 1]; m++) {
                const pgx_mem &mm = r.mems[m];
                out += "MEM START: " + std::to_string(mm.start) + ", MEM END: " + std::to_string(mm.end) + " BWT START: " +
                       std::to_string(mm.bwt_start) + " SIZE: " + std::to_string(mm.size) + "\n"; // :118
                out += "Number of unique positions: " + std::to_string(r.pos_offsets[m + 1] - r.pos_offsets[m]) + "\n"; // tag_arrays.cpp:885
                for (uint64_t p = r.pos_offsets[m]; p < r.pos_offsets[m + 1]; p++) out += std::to_string(r.positions[p]) + ", ";
                out += "\n";
            }
            out += "\n"; // :138
            if (out.size() > (1u << 22)) { std::cout << out; out.clear(); }
        }
        std::cout << out;
        std::cerr << err;
    }
    pgx_batch_free(b);
    std::cout << "\nTotal time for finding all MEMs: " << total_mem_time << " seconds" << std::endl; // :144
    std::cout << "Total time for all tag queries: " << total_tag_time << " seconds" << std::endl;   // :145
    pgx_index_close(h);
    return 0;
}

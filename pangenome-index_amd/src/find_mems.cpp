// find_mems -- drop-in for the reference CLI (src/find_mems.cpp:13-147) on MI355X.
//
//   find_mems <r_index.ri> <tags> <reads.txt> <min_mem_length> <min_occ> [options after the 5 positionals]
//
// stdout grammar is the reference's, byte for byte (find_mems.cpp:115-118,138,144-145 and
// tag_arrays.cpp:885-889); reads are processed in device batches but printed in file order.
// Options (ours): --device N, --mode compat|strict, --batch N (reads per device batch),
//                 --tags-format auto|bytecode|compact, --quiet (no per-read stderr line)
// The tag file may be either query format; the reference's find_mems only loads the sdsl-compact one.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pgx.h"

// decimal text of v at p; returns the end.  Two digits per division.
static inline char *put_u64(char *p, uint64_t v) {
    static const char d2[201] =
        "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    char tmp[20];
    int k = 20;
    while (v >= 100) {
        const unsigned r = (unsigned)(v % 100);
        v /= 100;
        k -= 2;
        tmp[k] = d2[2 * r];
        tmp[k + 1] = d2[2 * r + 1];
    }
    if (v >= 10) { k -= 2; tmp[k] = d2[2 * v]; tmp[k + 1] = d2[2 * v + 1]; }
    else tmp[--k] = (char)('0' + v);
    std::memcpy(p, tmp + k, (size_t)(20 - k));
    return p + (20 - k);
}
static inline char *put_str(char *p, const char *s) {
    const size_t n = std::strlen(s);
    std::memcpy(p, s, n);
    return p + n;
}

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
    int first_opt = 6;
    // find_mems_chunked.cpp:15-28 takes an optional sixth positional (chunk_size_mb of its memory-mapped loader): accepted, unused
    if (argc > 6 && argv[6][0] >= '0' && argv[6][0] <= '9') first_opt = 7;
    for (int i = first_opt; i < argc; i++) {
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

    std::ifstream reads(reads_file, std::ios::binary);
    if (!reads) { std::cerr << "Cannot open reads file: " << reads_file << std::endl; std::exit(EXIT_FAILURE); } // :91

    // The reference formats with iostreams one value at a time; a device batch finishes in milliseconds, so the text
    // side is what a user waits for: lines are split with memchr from 16 MiB chunks, and each batch is formatted by a
    // pool of threads (contiguous read ranges, one buffer each) and written in read order.
    const unsigned n_fmt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    pgx_batch *b = nullptr;
    std::string cat, carry;
    std::vector<uint64_t> offs;
    std::vector<char> chunk(16u << 20);
    size_t chunk_len = 0, chunk_pos = 0;
    size_t seq_no = 0;
    bool eof = false;
    // next line of the reads file (without the '\n') appended to `cat`; false at end of file
    auto next_line = [&](bool &empty) -> bool {
        for (;;) {
            if (chunk_pos < chunk_len) {
                const char *base = chunk.data() + chunk_pos;
                const char *nl = static_cast<const char *>(std::memchr(base, '\n', chunk_len - chunk_pos));
                if (nl) {
                    const size_t len = (size_t)(nl - base);
                    empty = carry.empty() && len == 0;
                    if (!carry.empty()) { cat += carry; carry.clear(); }
                    cat.append(base, len);
                    chunk_pos += len + 1;
                    return true;
                }
                carry.append(base, chunk_len - chunk_pos); // line continues in the next chunk
                chunk_pos = chunk_len;
            }
            if (eof) {
                if (carry.empty()) return false;
                empty = false; // std::getline returns a last line without a terminator
                cat += carry;
                carry.clear();
                return true;
            }
            reads.read(chunk.data(), (std::streamsize)chunk.size());
            chunk_len = (size_t)reads.gcount();
            chunk_pos = 0;
            if (chunk_len == 0) eof = true;
        }
    };
    bool done = false;
    std::vector<std::string> outs(n_fmt), errs(n_fmt);
    while (!done) {
        cat.clear();
        offs.assign(1, 0);
        while (offs.size() <= batch_reads) {
            bool empty = false;
            if (!next_line(empty)) { done = true; break; }
            if (empty) continue; // :97
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
        auto format_range = [&](unsigned w) {
            const size_t lo = n * w / n_fmt, hi = n * (w + 1) / n_fmt;
            std::string &out = outs[w], &err = errs[w];
            out.clear();
            err.clear();
            // exact upper bound of the text of this range: 20 digits per number
            const uint64_t m_lo = r.mem_offsets[lo], m_hi = r.mem_offsets[hi];
            const uint64_t p_cnt = r.pos_offsets[m_hi] - r.pos_offsets[m_lo];
            out.resize((hi - lo) * 32 + (m_hi - m_lo) * 192 + p_cnt * 22 + 64);
            char *p = &out[0];
            for (size_t i = lo; i < hi; i++) {
                const size_t seq = seq_no + i + 1;
                if (!quiet) { err += "[find_all_mems] total mems="; err += std::to_string(r.mem_offsets[i + 1] - r.mem_offsets[i]); err += '\n'; } // algorithm.hpp:754
                p = put_str(p, "Seq: "); p = put_u64(p, seq); *p++ = '\n'; // :115
                for (uint64_t m = r.mem_offsets[i]; m < r.mem_offsets[i + 1]; m++) {
                    const pgx_mem &mm = r.mems[m];
                    p = put_str(p, "MEM START: "); p = put_u64(p, mm.start);
                    p = put_str(p, ", MEM END: "); p = put_u64(p, mm.end);
                    p = put_str(p, " BWT START: "); p = put_u64(p, mm.bwt_start);
                    p = put_str(p, " SIZE: ");
                    if (mm.size < 0) { *p++ = '-'; p = put_u64(p, (uint64_t)(-(mm.size + 1)) + 1u); } else p = put_u64(p, (uint64_t)mm.size);
                    *p++ = '\n'; // :118
                    p = put_str(p, "Number of unique positions: "); p = put_u64(p, r.pos_offsets[m + 1] - r.pos_offsets[m]); *p++ = '\n'; // tag_arrays.cpp:885
                    for (uint64_t q = r.pos_offsets[m]; q < r.pos_offsets[m + 1]; q++) { p = put_u64(p, r.positions[q]); *p++ = ','; *p++ = ' '; }
                    *p++ = '\n';
                }
                *p++ = '\n'; // :138
            }
            out.resize((size_t)(p - &out[0]));
        };
        if (n_fmt == 1 || n < 4096) {
            for (unsigned w = 0; w < n_fmt; w++) format_range(w);
        } else {
            std::vector<std::thread> pool;
            for (unsigned w = 0; w < n_fmt; w++) pool.emplace_back(format_range, w);
            for (auto &th : pool) th.join();
        }
        for (unsigned w = 0; w < n_fmt; w++) {
            std::fwrite(outs[w].data(), 1, outs[w].size(), stdout);
            if (!errs[w].empty()) std::fwrite(errs[w].data(), 1, errs[w].size(), stderr);
        }
        seq_no += n;
    }
    std::fflush(stdout);
    pgx_batch_free(b);
    std::cout.flush();
    std::cout << "\nTotal time for finding all MEMs: " << total_mem_time << " seconds" << std::endl; // :144
    std::cout << "Total time for all tag queries: " << total_tag_time << " seconds" << std::endl;   // :145
    pgx_index_close(h);
    return 0;
}

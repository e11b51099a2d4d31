// query_tags -- drop-in for the reference CLI (src/query_tags.cpp:39-115) on MI355X.
//
//   query_tags <r_index.ri> <compressed_tags.tags> <reads.txt>     [--device N] [--mode compat|strict]
//
// Per non-empty line: FastLocate::count / count_encoded (backward search), then, when the range is not
// empty, TagArray::query_compressed_compact on it (prints "Number of unique positions: N" and the
// positions, src/tag_arrays.cpp:885-889) followed by the summary line of src/query_tags.cpp:103-108.
// All reads are searched in one device batch, all tag queries in a second one; output in file order.
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/pgx.h"

int main(int argc, char **argv) {
    if (argc < 4) {
        std::cerr << "Usage: ./bin/query_tags <r_index.ri> <compressed_tags.tags> <reads.txt>" << std::endl; // :41
        return 1;
    }
    int device = 0;
    uint32_t mode = PGX_MODE_COMPAT;
    for (int i = 4; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--device" && i + 1 < argc) device = std::stoi(argv[++i]);
        else if (a == "--mode" && i + 1 < argc) mode = std::string(argv[++i]) == "strict" ? PGX_MODE_STRICT : PGX_MODE_COMPAT;
        else { std::cerr << "unknown option " << a << std::endl; return 1; }
    }
    std::cerr << "Reading the rindex file" << std::endl;
    { std::ifstream rin(argv[1], std::ios::binary); if (!rin) { std::cerr << "Cannot open r-index: " << argv[1] << std::endl; return 1; } } // :56
    pgx_index *h = nullptr;
    std::cerr << "Reading the tag array index" << std::endl;
    if (pgx_index_open(argv[1], argv[2], PGX_TAGS_AUTO, mode, &h) != PGX_OK) { std::cerr << pgx_last_error() << std::endl; return 1; }
    pgx_index_info info;
    pgx_index_info_get(h, &info);
    std::vector<std::string> reads; // readSequencesFromFile, :24-35
    {
        std::ifstream file(argv[3]);
        std::string line;
        while (std::getline(file, line)) if (!line.empty()) reads.push_back(line);
    }
    if (reads.empty()) { std::cerr << "No reads found in file." << std::endl; return 1; } // :81-84
    std::string cat;
    std::vector<uint64_t> offs{0};
    for (auto &r : reads) { cat += r; offs.push_back(cat.size()); }
    std::vector<pgx_range> ranges(reads.size());
    if (pgx_count_batch(h, device, reinterpret_cast<const uint8_t *>(cat.data()), offs.data(), reads.size(), ranges.data()) != PGX_OK) {
        std::cerr << pgx_last_error() << std::endl;
        return 1;
    }
    std::vector<uint64_t> qs, qe, which;
    for (size_t i = 0; i < reads.size(); i++)
        if (ranges[i].first <= ranges[i].second) { qs.push_back(ranges[i].first); qe.push_back(ranges[i].second); which.push_back(i); }
    std::vector<uint64_t> rn(qs.size() ? qs.size() : 1), po(qs.size() + 1, 0), pos(1);
    uint64_t over = 0;
    if (!qs.empty()) {
        if (pgx_tag_query_batch(h, device, qs.data(), qe.data(), qs.size(), rn.data(), po.data(), nullptr, 0, &over) != PGX_OK) { std::cerr << pgx_last_error() << std::endl; return 1; }
        pos.resize(po.back() ? po.back() : 1);
        if (pgx_tag_query_batch(h, device, qs.data(), qe.data(), qs.size(), rn.data(), po.data(), pos.data(), pos.size(), &over) != PGX_OK) { std::cerr << pgx_last_error() << std::endl; return 1; }
    }
    size_t q = 0;
    for (size_t i = 0; i < reads.size(); i++) {
        if (info.is_encoded) { // :89-92
            std::cerr << "[encoded count] read_index=" << i << " len=" << reads[i].size() << std::endl;
            std::cerr << "  final range=[" << ranges[i].first << "," << ranges[i].second << "]\n";
        }
        if (ranges[i].first > ranges[i].second) { std::cerr << "Read " << i << " has no matches" << std::endl; continue; } // :97-100
        std::cout << "Number of unique positions: " << (po[q + 1] - po[q]) << "\n";
        for (uint64_t p = po[q]; p < po[q + 1]; p++) std::cout << pos[p] << ", ";
        std::cout << "\n";
        std::cout << "read_index=" << i << "\tlen=" << reads[i].size() << "\tbwt_start=" << ranges[i].first << "\tbwt_end=" << ranges[i].second
                  << "\truns=" << rn[q] << "\n"; // :103-108
        q++;
    }
    std::cout.flush();
    pgx_index_close(h);
    return 0;
}

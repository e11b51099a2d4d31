// merge_tags -- the reference CLI (src/merge_tags.cpp:434-869) on MI355X.
//
//   merge_tags <graph.gbz> <whole_genome.ri> <tag_dir> [--out FILE] [--device N]      the reference's argument list (:443-445)
//   merge_tags <seq_map.txt | --counts ...> <whole_genome.ri> <tag_dir> [...]          the sequence -> tag file map stated directly
//
// With a GBZ (recognised by its "GBZ " tag) the map is derived like the reference does: first node of path s -> weakly connected
// component -> the file of <tag_dir> whose first tag lies in that component (pgx_merge_tags_gbz; every regular file of the
// directory is a tag file, like get_files_in_dir :408-426).  Otherwise:
//
// seq_map.txt: one line per sequence of the whole-genome r-index, in sequence order: the name of the per-chromosome tag
// file (inside <tag_dir>) that holds the tags of that sequence.  "--counts f0:n0,f1:n1,..." replaces the file when the
// sequences are grouped (the first n0 sequences belong to f0, the next n1 to f1, ...), which is what concatenating
// per-chromosome texts gives.  Output: whole_genome_tag_array_compressed.tags (merge_tags.cpp:541), sdsl-compact format.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <sys/stat.h>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "../../include/pgx.h"

int main(int argc, char **argv) {
    if (argc < 4) {
        std::cerr << "usage: merge_tags <graph.gbz | seq_map.txt | --counts f0:n0,f1:n1,...> <whole_genome.ri> <tag_dir> [--out FILE] [--device N] [--reference-runs]" << std::endl;
        return EXIT_FAILURE;
    }
    {   // the reference's argv: <graph.gbz> <r_index> <tag_dir>
        char tag4[4] = {0, 0, 0, 0};
        std::ifstream probe(argv[1], std::ios::binary);
        if (probe && probe.read(tag4, 4) && std::memcmp(tag4, "GBZ ", 4) == 0) {
            const std::string ri = argv[2], dir = argv[3];
            std::string out = "whole_genome_tag_array_compressed.tags"; // written into the working directory (:538)
            int device = 0;
            uint32_t flags = 0;
            for (int i = 4; i < argc; i++) {
                const std::string o = argv[i];
                if (o == "--out" && i + 1 < argc) out = argv[++i];
                else if (o == "--device" && i + 1 < argc) device = std::stoi(argv[++i]);
                else if (o == "--reference-runs") flags |= PGX_MERGE_REFERENCE_RUNS;
                else { std::cerr << "unknown option " << o << std::endl; return EXIT_FAILURE; }
            }
            std::cerr << "Loading the graph file" << std::endl;            // :449
            std::cerr << "Getting the lists of tag files" << std::endl;    // :452
            std::vector<std::string> files;
            if (DIR *d = opendir(dir.c_str())) {
                while (dirent *e = readdir(d)) {
                    const std::string p = dir + "/" + e->d_name;
                    struct stat st;
                    if (stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode)) files.push_back(p);
                }
                closedir(d);
            } else { std::cerr << "Cannot open tag directory: " << dir << std::endl; return EXIT_FAILURE; }
            std::sort(files.begin(), files.end());
            std::cerr << "The list of files are: " << std::endl;          // :459
            std::vector<const char *> cp;
            for (auto &p : files) { std::cerr << p << std::endl; cp.push_back(p.c_str()); }
            std::cerr << "Reading the whole genome r-index file (encoded)" << std::endl; // :465
            std::cerr << "Finding the node to component mapping" << std::endl;           // :477
            if (pgx_merge_tags_gbz_ex(argv[1], ri.c_str(), cp.data(), (uint32_t)cp.size(), device, out.c_str(), flags) != PGX_OK) {
                std::cerr << pgx_last_error() << std::endl;
                return EXIT_FAILURE;
            }
            std::cerr << "Index files merged and ready to use!" << std::endl; // :855
            return 0;
        }
    }
    int a = 1;
    std::vector<std::string> seq_file; // per sequence: tag file name
    if (std::string(argv[a]) == "--counts") {
        if (argc < 5) { std::cerr << "missing value for --counts" << std::endl; return EXIT_FAILURE; }
        std::string spec = argv[a + 1];
        a += 2;
        size_t p = 0;
        while (p < spec.size()) {
            size_t q = spec.find(',', p);
            if (q == std::string::npos) q = spec.size();
            const std::string item = spec.substr(p, q - p);
            const size_t c = item.rfind(':');
            if (c == std::string::npos) { std::cerr << "bad --counts item: " << item << std::endl; return EXIT_FAILURE; }
            const size_t cnt = (size_t)std::stoull(item.substr(c + 1));
            seq_file.insert(seq_file.end(), cnt, item.substr(0, c));
            p = q + 1;
        }
    } else {
        std::ifstream in(argv[a]);
        if (!in) { std::cerr << "Cannot open sequence map: " << argv[a] << std::endl; return EXIT_FAILURE; }
        std::string line;
        while (std::getline(in, line))
            if (!line.empty()) seq_file.push_back(line);
        a += 1;
    }
    if (argc < a + 2) { std::cerr << "missing <whole_genome.ri> <tag_dir>" << std::endl; return EXIT_FAILURE; }
    const std::string ri = argv[a], dir = argv[a + 1];
    std::string out = "whole_genome_tag_array_compressed.tags";
    int device = 0;
    uint32_t flags = 0;
    for (int i = a + 2; i < argc; i++) {
        const std::string o = argv[i];
        if (o == "--out" && i + 1 < argc) out = argv[++i];
        else if (o == "--device" && i + 1 < argc) device = std::stoi(argv[++i]);
        else if (o == "--reference-runs") flags |= PGX_MERGE_REFERENCE_RUNS;
        else { std::cerr << "unknown option " << o << std::endl; return EXIT_FAILURE; }
    }
    std::map<std::string, uint32_t> id;
    std::vector<std::string> paths;
    std::vector<uint32_t> s2f(seq_file.size());
    for (size_t s = 0; s < seq_file.size(); s++) {
        auto it = id.find(seq_file[s]);
        if (it == id.end()) {
            it = id.emplace(seq_file[s], (uint32_t)paths.size()).first;
            paths.push_back(dir + "/" + seq_file[s]);
        }
        s2f[s] = it->second;
    }
    std::cerr << "The list of files are: " << std::endl; // merge_tags.cpp:459
    std::vector<const char *> cp;
    for (auto &p : paths) { std::cerr << p << std::endl; cp.push_back(p.c_str()); }
    std::cerr << "Merging tags and creating the whole genome tag array indexing" << std::endl; // :735
    if (pgx_merge_tags_ex(ri.c_str(), cp.data(), (uint32_t)cp.size(), s2f.data(), s2f.size(), device, out.c_str(), flags) != PGX_OK) {
        std::cerr << pgx_last_error() << std::endl;
        return EXIT_FAILURE;
    }
    std::cerr << "Index files merged and ready to use!" << std::endl; // :855
    return 0;
}

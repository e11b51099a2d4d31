// find_mems -- drop-in for the reference CLI (src/find_mems.cpp:13-147) on MI355X.
//
//   find_mems <r_index.ri> <tags> <reads.txt> <min_mem_length> <min_occ> [options after the 5 positionals]
//
// stdout grammar is the reference's, byte for byte (find_mems.cpp:115-118,138,144-145 and
// tag_arrays.cpp:885-889); reads are processed in device batches but printed in file order.
// Options (ours): --device N | --gpus N (devices 0 .. N-1) | --devices a,b,.. ; --streams W (batches in flight per device);
//                 --mode compat|strict, --batch N (reads per device batch), --tags-format auto|bytecode|compact,
//                 --quiet (no per-read stderr line)
// The tag file may be either query format; the reference's find_mems only loads the sdsl-compact one.
//
// The per-read loop of the reference (find_mems.cpp:94-139) becomes a pipeline: one reader thread cuts the reads file into
// batches of consecutive reads; every device has W worker threads, each with its own batch and stream (pgx_batch), which
// upload, run, download and format the text of whole batches -- so on one device the upload of batch k + 1, the kernels of
// batch k and the download / formatting of batch k - 1 overlap, and N devices take batches in turn with the index replicated
// (reads shard, no collective); the main thread writes the finished texts in file order.
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
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
#include <atomic>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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

// read bytes of a batch: pinned host memory from a small pool (pgx_host_alloc: uploads at link speed), ordinary memory when that fails
struct ReadBuf {
    char *p = nullptr;
    size_t cap = 0, len = 0;
    bool pinned = false;
};
class ReadBufPool {
    std::mutex mu;
    std::vector<ReadBuf> free_;
public:
    ReadBuf get(size_t bytes) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].cap >= bytes) { ReadBuf b = free_[i]; free_.erase(free_.begin() + (std::ptrdiff_t)i); b.len = 0; return b; }
        }
        ReadBuf b;
        b.cap = bytes + bytes / 8 + 4096;
        void *q = nullptr;
        if (pgx_host_alloc(b.cap, &q) == PGX_OK) { b.p = static_cast<char *>(q); b.pinned = true; }
        else { b.p = static_cast<char *>(std::malloc(b.cap)); if (!b.p) throw std::bad_alloc(); }
        return b;
    }
    void put(ReadBuf b) { if (!b.p) return; std::lock_guard<std::mutex> lk(mu); free_.push_back(b); }
    ~ReadBufPool() { for (auto &b : free_) { if (b.pinned) pgx_host_free(b.p); else std::free(b.p); } }
};
struct Job { // one batch of consecutive reads
    uint64_t id = 0, first_seq = 0; // batch number, reads before it in the file
    ReadBuf cat;
    std::vector<uint64_t> offs;
    // the same reads packed to two bits per symbol by the parse thread (pgx_pack_reads): [packed words | bytes of the reads with a byte outside A C G T]
    ReadBuf pk;
    std::vector<uint64_t> side_ids;
    uint64_t n_side = 0, side_at = 0;
    bool packed = false;
};
struct Done {
    std::vector<std::string> outs, errs; // text pieces in read order
    double mem_s = 0, tag_s = 0;
    std::string error;
};

// text of reads [lo, hi) of a finished batch (stdout piece + the per-read stderr lines of find_all_mems, algorithm.hpp:754)
static void format_range(const pgx_result &r, size_t lo, size_t hi, uint64_t first_seq, bool quiet, std::string &out, std::string &err) {
    out.clear();
    err.clear();
    // exact upper bound of the text of this range: 20 digits per number
    const uint64_t m_lo = r.mem_offsets[lo], m_hi = r.mem_offsets[hi];
    const uint64_t p_cnt = r.pos_offsets[m_hi] - r.pos_offsets[m_lo];
    out.resize((hi - lo) * 32 + (m_hi - m_lo) * 192 + p_cnt * 22 + 64);
    char *p = &out[0];
    for (size_t i = lo; i < hi; i++) {
        const size_t seq = first_seq + i + 1;
        if (!quiet) { err += "[find_all_mems] total mems="; err += std::to_string(r.mem_offsets[i + 1] - r.mem_offsets[i]); err += '\n'; }
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
}

int main(int argc, char **argv) {
    if (argc < 6) {
        std::cerr << "usage: find_mems <r_index.ri> <tags> <reads.txt> <min_mem_length> <min_occ>"
                     " [--device N | --gpus N | --devices a,b,..] [--streams W] [--mode compat|strict] [--batch N]"
                     " [--tags-format auto|bytecode|compact] [--quiet]" << std::endl;
        return EXIT_FAILURE;
    }
    const std::string r_index_file = argv[1], tag_array_index = argv[2], reads_file = argv[3];
    const size_t mem_length = (size_t)std::stoi(argv[4]); // find_mems.cpp:17 (std::stoi -> size_t)
    const size_t min_occ = (size_t)std::stoi(argv[5]);
    std::vector<int> devices(1, 0);
    uint32_t mode = PGX_MODE_COMPAT, tfmt = PGX_TAGS_AUTO;
    size_t batch_reads = 1u << 18; // (reads per batch.  The device alone would like 2^20 -- fresh batches 177 M reads/s at 2^18, 278 M at 2^20, profiles/r04_batch_size_sweep.txt --, but this program is bound by its own host stages: 16 M reads take 0.5-0.7 s with 2^18 and 2.7-2.9 s with 2^20, pinned buffers of 150 MB per job and the formatting of the first and last batch: profiles/r04_cli_e2e.txt)
    unsigned streams = 3;
    bool quiet = false;
    int first_opt = 6;
    // find_mems_chunked.cpp:15-28 takes an optional sixth positional (chunk_size_mb of its memory-mapped loader): accepted, unused
    if (argc > 6 && argv[6][0] >= '0' && argv[6][0] <= '9') first_opt = 7;
    for (int i = first_opt; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; std::exit(EXIT_FAILURE); } return argv[++i]; };
        if (a == "--device") devices.assign(1, std::stoi(next()));
        else if (a == "--gpus") { const int g = std::stoi(next()); devices.clear(); for (int d = 0; d < std::max(1, g); d++) devices.push_back(d); }
        else if (a == "--devices") {
            devices.clear();
            std::stringstream ss(next());
            for (std::string tok; std::getline(ss, tok, ',');) if (!tok.empty()) devices.push_back(std::stoi(tok));
            if (devices.empty()) { std::cerr << "--devices: empty list" << std::endl; return EXIT_FAILURE; }
        }
        else if (a == "--streams") streams = (unsigned)std::max(1, std::stoi(next()));
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
    for (int d : devices) // the index is replicated: one image per distinct device
        if (pgx_index_to_device(h, d) != PGX_OK) { std::cerr << pgx_last_error() << std::endl; return EXIT_FAILURE; }
    auto time3 = std::chrono::high_resolution_clock::now();
    std::cerr << "Loading tag arrays took " << std::chrono::duration<double>(time3 - time2).count() << " seconds" << std::endl;

    // ---- the reads file, mapped: the reference reads it with std::getline (find_mems.cpp:96); here a planner cuts it into byte ranges that
    //      end at a newline, PARSE threads turn ranges into batches side by side (line splitting was what bounded the single reader thread of
    //      round 2 at ~1 GB/s), device workers take the batches in file order ----
    struct MappedFile {
        const char *p = nullptr;
        size_t n = 0;
        std::vector<char> own; // fallback when the file cannot be mapped (a pipe, a FIFO)
        int fd = -1;
        bool mapped = false;
        ~MappedFile() { if (mapped) ::munmap(const_cast<char *>(p), n); if (fd >= 0) ::close(fd); }
    } mf;
    {
        mf.fd = ::open(reads_file.c_str(), O_RDONLY);
        if (mf.fd < 0) { std::cerr << "Cannot open reads file: " << reads_file << std::endl; std::exit(EXIT_FAILURE); } // :91
        struct stat st;
        if (::fstat(mf.fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, mf.fd, 0);
            if (m != MAP_FAILED) { mf.p = static_cast<const char *>(m); mf.n = (size_t)st.st_size; mf.mapped = true; (void)::madvise(m, mf.n, MADV_SEQUENTIAL); }
        }
        if (!mf.mapped) { // read it all (what the mapping would have paged in)
            char buf[1 << 16];
            for (;;) { const ssize_t k = ::read(mf.fd, buf, sizeof buf); if (k <= 0) break; mf.own.insert(mf.own.end(), buf, buf + k); }
            mf.p = mf.own.data(); mf.n = mf.own.size();
        }
    }

    // ---- queues ----
    const unsigned n_workers = (unsigned)devices.size() * streams;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n_fmt = std::max(1u, std::min(16u, hw / n_workers)); // formatting threads per batch
    unsigned n_parse = std::max(1u, std::min(8u, hw / 4));
    if (const char *e = std::getenv("PGX_CLI_PARSE_THREADS")) n_parse = (unsigned)std::max(1, std::atoi(e));
    std::mutex mu;
    std::condition_variable cv_jobs, cv_done, cv_space;
    std::map<uint64_t, std::unique_ptr<Job>> parsed;     // by id, waiting for a device worker
    std::map<uint64_t, std::unique_ptr<Done>> finished;
    bool reader_done = false, failed = false;
    uint64_t n_jobs = 0, next_write = 0, next_take = 0, next_seq_id = 0, seq_so_far = 0;
    std::map<uint64_t, uint64_t> counts;                  // reads of parsed ranges whose first_seq is not known yet
    const size_t max_ahead = 2 * n_workers + n_parse + 2; // ranges parsed or finished but not yet written (bounds memory)

    // ranges: about batch_reads reads each, by the length of the first lines
    std::vector<std::pair<size_t, size_t>> ranges;
    {
        size_t probe = std::min<size_t>(mf.n, 1u << 16), lines = 0;
        for (size_t i = 0; i < probe; i++) lines += mf.p[i] == '\n';
        const double per_line = lines ? (double)probe / (double)lines : 152.0;
        const size_t want = (size_t)std::max(1.0, per_line * (double)batch_reads);
        size_t at = 0;
        while (at < mf.n) {
            size_t end = std::min(mf.n, at + want);
            if (end < mf.n) {
                const char *nl = static_cast<const char *>(std::memchr(mf.p + end, '\n', mf.n - end));
                end = nl ? (size_t)(nl - mf.p) + 1 : mf.n;
            }
            ranges.emplace_back(at, end);
            at = end;
        }
    }
    n_jobs = ranges.size();
    std::atomic<uint64_t> next_range{0};

    // a range -> the reads it holds, concatenated, with offsets; empty lines are skipped (:97); a last line without a newline counts (std::getline)
    ReadBufPool pool;
    // PGX_CLI_PACKED=1: the parse threads also pack the reads to two bits per symbol (pgx_pack_reads) and the batch travels packed (pgx_batch_upload_packed).
    // Off by default: this program is bound by its host stages, not by the link -- 16 M reads, three device workers, same box: pipeline 0.65 s with byte
    // uploads, 0.88 s with packed ones (the extra pass over the bytes in the parse threads: profiles/r04_cli_e2e.txt).  A caller whose batches are already
    // packed, or whose pipeline is bound by the device, gains: bench.py fresh_batch 366 against 220 M reads/s.
    const bool use_packed = std::getenv("PGX_CLI_PACKED") && std::getenv("PGX_CLI_PACKED")[0] == '1';
    auto parse_range = [&](uint64_t id) {
        std::unique_ptr<Job> j(new Job());
        const char *q = mf.p + ranges[id].first, *end = mf.p + ranges[id].second;
        j->id = id;
        j->cat = pool.get((size_t)(end - q) + 1);
        j->offs.reserve((size_t)(end - q) / 100 + 16);
        j->offs.push_back(0);
        char *dst = j->cat.p;
        while (q < end) {
            const char *nl = static_cast<const char *>(std::memchr(q, '\n', (size_t)(end - q)));
            const size_t len = nl ? (size_t)(nl - q) : (size_t)(end - q);
            if (len) { std::memcpy(dst, q, len); dst += len; j->offs.push_back((uint64_t)(dst - j->cat.p)); }
            q += len + 1;
        }
        j->cat.len = (size_t)(dst - j->cat.p);
        // packed for the upload: a quarter of the bytes over the link and no pass over them on the device (PGX_CLI_PACKED=0: bytes as they are)
        if (use_packed && j->offs.size() > 1) {
            const size_t words = (j->cat.len + 15) / 16, side_cap = j->cat.len / 8 + 4096;
            j->pk = pool.get(words * 4 + side_cap);
            j->side_at = words * 4;
            j->side_ids.resize(side_cap / 64 + 64);
            uint64_t n_side = 0, n_side_bytes = 0;
            const pgx_status st = pgx_pack_reads(reinterpret_cast<const uint8_t *>(j->cat.p), j->offs.data(), j->offs.size() - 1, 1, reinterpret_cast<uint32_t *>(j->pk.p),
                                                 j->side_ids.data(), j->side_ids.size(), reinterpret_cast<uint8_t *>(j->pk.p) + j->side_at, side_cap, &n_side, &n_side_bytes);
            if (st == PGX_OK) { j->packed = true; j->n_side = n_side; }
            else { pool.put(j->pk); j->pk = ReadBuf(); } // (too many such reads for the side list: this batch travels as bytes)
        }
        return j;
    };
    // PGX_CLI_STATS=1: busy seconds of every stage (summed over its threads) on stderr at the end
    const bool cli_stats = std::getenv("PGX_CLI_STATS") != nullptr;
    std::atomic<uint64_t> ns_parse{0}, ns_upload{0}, ns_run{0}, ns_download{0}, ns_format{0}, ns_write{0};
    auto now_ns = []() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const uint64_t t_pipeline0 = now_ns();
    std::vector<std::thread> parsers;
    for (unsigned t = 0; t < n_parse; t++)
        parsers.emplace_back([&]() {
            for (;;) {
                const uint64_t id = next_range.fetch_add(1);
                if (id >= n_jobs) break;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_space.wait(lk, [&]() { return failed || id < next_write + max_ahead; });
                    if (failed) break;
                }
                const uint64_t tp0 = now_ns();
                std::unique_ptr<Job> j = parse_range(id);
                ns_parse += now_ns() - tp0;
                std::lock_guard<std::mutex> lk(mu);
                counts[id] = j->offs.size() - 1;
                parsed[id] = std::move(j);
                // sequence numbers: a range knows its first read's number once every range before it has been parsed
                while (counts.count(next_seq_id)) {
                    parsed[next_seq_id]->first_seq = seq_so_far;
                    seq_so_far += counts[next_seq_id];
                    counts.erase(next_seq_id);
                    next_seq_id++;
                }
                cv_jobs.notify_all();
            }
        });
    if (n_jobs == 0) reader_done = true;

    auto worker = [&](int device) {
        pgx_batch *b = nullptr; // long-lived: its device and pinned host buffers are reused by every batch of this worker
        for (;;) {
            std::unique_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                // the next range in file order, once it is parsed and numbered
                cv_jobs.wait(lk, [&]() { return failed || next_take >= n_jobs || (parsed.count(next_take) && next_take < next_seq_id); });
                if (failed || next_take >= n_jobs) break;
                j = std::move(parsed[next_take]);
                parsed.erase(next_take);
                next_take++;
                if (next_take >= n_jobs) { reader_done = true; cv_jobs.notify_all(); cv_done.notify_all(); }
            }
            std::unique_ptr<Done> d(new Done());
            const size_t n = j->offs.size() - 1;
            if (n == 0) { // a range of empty lines only
                pool.put(j->cat);
                std::lock_guard<std::mutex> lk(mu);
                finished[j->id] = std::move(d);
                cv_done.notify_all();
                continue;
            }
            pgx_result r;
            const uint8_t *rp = reinterpret_cast<const uint8_t *>(j->cat.p);
            uint64_t t0 = now_ns();
            pgx_status st = PGX_OK;
            if (j->packed) {
                static const uint64_t no_offsets[1] = {0};
                if (!b) st = pgx_batch_create(h, device, nullptr, no_offsets, 0, &b);
                if (st == PGX_OK)
                    st = pgx_batch_upload_packed(b, reinterpret_cast<const uint32_t *>(j->pk.p), j->offs.data(), n, j->side_ids.data(),
                                                 reinterpret_cast<const uint8_t *>(j->pk.p) + j->side_at, j->n_side);
            } else st = b ? pgx_batch_upload(b, rp, j->offs.data(), n) : pgx_batch_create(h, device, rp, j->offs.data(), n, &b);
            uint64_t t1 = now_ns();
            if (st == PGX_OK) st = pgx_batch_run(b, mem_length, min_occ, PGX_RUN_TAGS | PGX_RUN_TIMING, nullptr);
            uint64_t t2 = now_ns();
            if (st == PGX_OK) st = pgx_batch_result(b, &r);
            uint64_t t3 = now_ns();
            ns_upload += t1 - t0; ns_run += t2 - t1; ns_download += t3 - t2;
            pool.put(j->cat); // (the upload has completed: pgx_batch_upload synchronises its copies)
            j->cat = ReadBuf();
            pool.put(j->pk);
            j->pk = ReadBuf();
            if (st != PGX_OK) d->error = pgx_last_error();
            else {
                pgx_timing t;
                if (pgx_batch_timing(b, &t) == PGX_OK) {
                    d->mem_s = 1e-3 * (t.ms_find_mems + t.ms_compact);
                    d->tag_s = 1e-3 * (t.ms_tag_locate + t.ms_tag_gather + t.ms_tag_sort);
                }
                const unsigned parts = n < 4096 ? 1u : n_fmt;
                d->outs.resize(parts);
                d->errs.resize(parts);
                if (parts == 1) format_range(r, 0, n, j->first_seq, quiet, d->outs[0], d->errs[0]);
                else {
                    std::vector<std::thread> pool;
                    for (unsigned w = 0; w < parts; w++)
                        pool.emplace_back([&, w]() { format_range(r, n * w / parts, n * (w + 1) / parts, j->first_seq, quiet, d->outs[w], d->errs[w]); });
                    for (auto &th : pool) th.join();
                }
                ns_format += now_ns() - t3;
            }
            std::lock_guard<std::mutex> lk(mu);
            if (!d->error.empty()) failed = true;
            finished[j->id] = std::move(d);
            cv_done.notify_all();
            if (failed) { cv_jobs.notify_all(); cv_space.notify_all(); }
        }
        pgx_batch_free(b);
    };
    std::vector<std::thread> workers;
    for (int d : devices)
        for (unsigned w = 0; w < streams; w++) workers.emplace_back(worker, d);

    // ---- writer: batches in file order ----
    std::string error;
    for (;;) {
        std::unique_ptr<Done> d;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&]() { return finished.count(next_write) || (reader_done && next_write == n_jobs) || (failed && !finished.count(next_write)); });
            if (!finished.count(next_write)) break;
            d = std::move(finished[next_write]);
            finished.erase(next_write);
            next_write++;
            cv_space.notify_all();
        }
        if (!d->error.empty()) { error = d->error; break; }
        const uint64_t tw0 = now_ns();
        for (size_t w = 0; w < d->outs.size(); w++) {
            std::fwrite(d->outs[w].data(), 1, d->outs[w].size(), stdout);
            if (!d->errs[w].empty()) std::fwrite(d->errs[w].data(), 1, d->errs[w].size(), stderr);
        }
        ns_write += now_ns() - tw0;
        total_mem_time += d->mem_s;
        total_tag_time += d->tag_s;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!error.empty()) failed = true;
        cv_jobs.notify_all();
        cv_space.notify_all();
    }
    for (auto &th : parsers) th.join();
    for (auto &th : workers) th.join();
    if (error.empty()) // a batch that failed behind one still in flight when the writer stopped
        for (auto &kv : finished)
            if (!kv.second->error.empty()) { error = kv.second->error; break; }
    if (error.empty() && failed) error = "find_mems: a device batch failed";
    if (!error.empty()) {
        std::cerr << error << std::endl;
        return EXIT_FAILURE;
    }
    std::fflush(stdout);
    std::cout.flush();
    if (cli_stats)
        std::fprintf(stderr, "[find_mems] pipeline %.3f s wall; busy seconds: parse %.3f (%u threads), upload %.3f, run %.3f, download %.3f, format %.3f (wall of %u-thread pools), write %.3f; %u device workers\n",
                     1e-9 * (double)(now_ns() - t_pipeline0), 1e-9 * (double)ns_parse.load(), n_parse, 1e-9 * (double)ns_upload.load(), 1e-9 * (double)ns_run.load(),
                     1e-9 * (double)ns_download.load(), 1e-9 * (double)ns_format.load(), n_fmt, 1e-9 * (double)ns_write.load(), n_workers);
    std::cout << "\nTotal time for finding all MEMs: " << total_mem_time << " seconds" << std::endl; // :144
    std::cout << "Total time for all tag queries: " << total_tag_time << " seconds" << std::endl;   // :145
    pgx_index_close(h);
    return 0;
}

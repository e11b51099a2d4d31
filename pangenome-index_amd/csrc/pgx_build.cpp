// pgx_build.cpp -- build-side helpers (CPU, run once): text -> RLBWT, RLBWT -> .ri, compact tag writer.
//
// These exist because no encoded .ri and no sdsl-compact tag file ships with the reference
// (SURVEY section 0 "Fixture reality check"): the find_mems path needs both as inputs.
//   pgx_build_rlbwt        what grlBWT produces for the reference (bwt_buff_reader layout)
//   pgx_build_rindex       FastLocate(std::string) src/r-index.cpp:778-965 (block building) +
//                          serialize_encoded :297-376 / serialize :266-294
//   pgx_write_compact_tags append_compact_run_streamed + merge_compressed_files_sdsl,
//                          src/tag_arrays.cpp:940-974, 622-654
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>

#include "pgx_host.hpp"

using namespace pgx;

#define PGX_GUARD_BEGIN try {
#define PGX_GUARD_END                                                                               \
    }                                                                                               \
    catch (const pgx::Error &e) { pgx::set_last_error(e.what()); return e.code; }                   \
    catch (const std::bad_alloc &) { pgx::set_last_error("out of host memory"); return PGX_ERR_NOMEM; } \
    catch (const std::exception &e) { pgx::set_last_error(e.what()); return PGX_ERR_FORMAT; }

// PGX_BUILD_TIMING=1: phase times of the builders on stderr
struct BuildTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = std::getenv("PGX_BUILD_TIMING") != nullptr;
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pgx build] %-28s %8.2f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

// sort with host threads: slices sorted side by side, then merged pairwise (std::inplace_merge), round by round
// Sort by a 64-bit key that is spread over [0, key_end): the elements are dealt into one bucket per thread by key range (count, scatter), the
// buckets sorted side by side -- no merge passes (the last merge of parallel_sort below runs on one thread: 65 M samples took 22 s on 8 cores).
template <class T, class Key> static void parallel_range_sort(std::vector<T> &v, uint64_t key_end, Key key) {
    const size_t n = v.size();
    const unsigned P = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 16u));
    auto less = [&](const T &a, const T &b) { return key(a) < key(b); };
    if (n < (1u << 20) || P < 2 || key_end == 0 || std::getenv("PGX_BUILD_SERIAL_SORT")) { std::sort(v.begin(), v.end(), less); return; } // (the variable: a check of this function)
    const unsigned B = 4 * P; // more buckets than threads: uneven key ranges even out
    auto bucket_of = [&](const T &x) { const unsigned __int128 t = (unsigned __int128)std::min<uint64_t>(key(x), key_end - 1) * B; return (unsigned)(t / key_end); };
    std::vector<std::vector<size_t>> cnt(P, std::vector<size_t>(B, 0));
    auto run = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < P; t++) th.emplace_back([&, t]() { fn(t, n * t / P, n * (t + 1) / P); });
        for (auto &x : th) x.join();
    };
    run([&](unsigned t, size_t a, size_t b) { for (size_t i = a; i < b; i++) cnt[t][bucket_of(v[i])]++; });
    std::vector<size_t> start(B + 1, 0);
    for (unsigned b = 0; b < B; b++) { size_t c = 0; for (unsigned t = 0; t < P; t++) c += cnt[t][b]; start[b + 1] = start[b] + c; }
    std::vector<std::vector<size_t>> at(P, std::vector<size_t>(B, 0));
    for (unsigned b = 0; b < B; b++) { size_t o = start[b]; for (unsigned t = 0; t < P; t++) { at[t][b] = o; o += cnt[t][b]; } }
    std::vector<T> out(n);
    run([&](unsigned t, size_t a, size_t b) { for (size_t i = a; i < b; i++) out[at[t][bucket_of(v[i])]++] = v[i]; });
    std::atomic<unsigned> next{0};
    run([&](unsigned, size_t, size_t) { for (;;) { const unsigned b = next.fetch_add(1); if (b >= B) break; std::sort(out.begin() + (std::ptrdiff_t)start[b], out.begin() + (std::ptrdiff_t)start[b + 1], less); } });
    v.swap(out);
}

template <class It, class Cmp> static void parallel_sort(It first, It last, Cmp cmp) {
    const size_t n = (size_t)(last - first);
    unsigned T = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 16u));
    if (n < (1u << 20) || T < 2) { std::sort(first, last, cmp); return; }
    unsigned P = 1;
    while (P * 2 <= T) P *= 2; // a power of two slices
    std::vector<size_t> cut(P + 1);
    for (unsigned i = 0; i <= P; i++) cut[i] = n * i / P;
    {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < P; i++) th.emplace_back([&, i]() { std::sort(first + (std::ptrdiff_t)cut[i], first + (std::ptrdiff_t)cut[i + 1], cmp); });
        for (auto &t : th) t.join();
    }
    for (unsigned w = 1; w < P; w *= 2) {
        std::vector<std::thread> th;
        for (unsigned i = 0; i + w < P; i += 2 * w)
            th.emplace_back([&, i, w]() {
                std::inplace_merge(first + (std::ptrdiff_t)cut[i], first + (std::ptrdiff_t)cut[i + w], first + (std::ptrdiff_t)cut[std::min(i + 2 * w, P)], cmp);
            });
        for (auto &t : th) t.join();
    }
}

template <class T> static void put(std::vector<uint8_t> &out, T v) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&v);
    out.insert(out.end(), p, p + sizeof(T));
}
static inline unsigned hi_bit(uint64_t x) { return x ? 63u - (unsigned)__builtin_clzll(x) : 0u; }

// ------------------------------------------------------------------------------------------
// SA-IS (Nong, Zhang, Chan 2009) over an int32 string whose last symbol is a unique minimum.
namespace {
struct Sais {
    static inline bool tget(const std::vector<uint8_t> &t, int64_t i) { return (t[(size_t)i >> 3] >> (i & 7)) & 1; }
    static inline void tset(std::vector<uint8_t> &t, int64_t i, bool b) {
        if (b) t[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7));
        else t[(size_t)i >> 3] &= (uint8_t)~(1u << (i & 7));
    }
    static inline bool is_lms(const std::vector<uint8_t> &t, int64_t i) { return i > 0 && tget(t, i) && !tget(t, i - 1); }

    static void buckets(const int32_t *s, int32_t n, int32_t K, std::vector<int32_t> &bkt, bool end) {
        std::fill(bkt.begin(), bkt.end(), 0);
        for (int32_t i = 0; i < n; i++) bkt[(size_t)s[i]]++;
        int32_t sum = 0;
        for (int32_t c = 0; c < K; c++) {
            sum += bkt[(size_t)c];
            bkt[(size_t)c] = end ? sum : sum - bkt[(size_t)c];
        }
    }
    static void induce(const std::vector<uint8_t> &t, int32_t *SA, const int32_t *s, std::vector<int32_t> &bkt, int32_t n, int32_t K) {
        buckets(s, n, K, bkt, false);
        for (int32_t i = 0; i < n; i++) {
            int32_t j = SA[i] - 1;
            if (SA[i] > 0 && !tget(t, j)) SA[bkt[(size_t)s[j]]++] = j;
        }
        buckets(s, n, K, bkt, true);
        for (int32_t i = n - 1; i >= 0; i--) {
            int32_t j = SA[i] - 1;
            if (SA[i] > 0 && tget(t, j)) SA[--bkt[(size_t)s[j]]] = j;
        }
    }
    static void run(const int32_t *s, int32_t *SA, int32_t n, int32_t K) {
        if (n == 1) { SA[0] = 0; return; }
        std::vector<uint8_t> t((size_t)n / 8 + 1, 0);
        tset(t, n - 1, true);
        for (int32_t i = n - 2; i >= 0; i--) tset(t, i, s[i] < s[i + 1] || (s[i] == s[i + 1] && tget(t, i + 1)));
        std::vector<int32_t> bkt((size_t)K);
        buckets(s, n, K, bkt, true);
        for (int32_t i = 0; i < n; i++) SA[i] = -1;
        for (int32_t i = 1; i < n; i++)
            if (is_lms(t, i)) SA[--bkt[(size_t)s[i]]] = i;
        induce(t, SA, s, bkt, n, K);
        int32_t n1 = 0;
        for (int32_t i = 0; i < n; i++)
            if (is_lms(t, SA[i])) SA[n1++] = SA[i];
        for (int32_t i = n1; i < n; i++) SA[i] = -1;
        int32_t name = 0, prev = -1;
        for (int32_t i = 0; i < n1; i++) {
            int32_t pos = SA[i];
            bool diff = false;
            for (int32_t d = 0; d < n; d++) {
                if (prev == -1 || s[pos + d] != s[prev + d] || tget(t, pos + d) != tget(t, prev + d)) { diff = true; break; }
                else if (d > 0 && (is_lms(t, pos + d) || is_lms(t, prev + d))) break;
            }
            if (diff) { name++; prev = pos; }
            SA[n1 + pos / 2] = name - 1;
        }
        for (int32_t i = n - 1, j = n - 1; i >= n1; i--)
            if (SA[i] >= 0) SA[j--] = SA[i];
        int32_t *SA1 = SA, *s1 = SA + n - n1;
        if (name < n1) run(s1, SA1, n1, name);
        else for (int32_t i = 0; i < n1; i++) SA1[s1[i]] = i;
        buckets(s, n, K, bkt, true);
        for (int32_t i = 1, j = 0; i < n; i++)
            if (is_lms(t, i)) s1[j++] = i;
        for (int32_t i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
        for (int32_t i = n1; i < n; i++) SA[i] = -1;
        for (int32_t i = n1 - 1; i >= 0; i--) {
            int32_t j = SA[i];
            SA[i] = -1;
            SA[--bkt[(size_t)s[j]]] = j;
        }
        induce(t, SA, s, bkt, n, K);
    }
};
} // namespace

// BWT of the collection S_0 \n S_1 \n ... with endmarkers ordered by sequence number (the
// convention of the reference's fixtures and of tests/test_rindex.cpp:35-60: rotations of the text
// with distinct increasing terminators).  Output: grlBWT run file (u64 bytes/symbol = 1,
// u64 bytes/length, then (symbol, length) records).
// The BWT of a collection as the builders below hand it on: grlBWT-style maximal runs, and -- for the SA samples of the .ri -- the text
// position of the suffix at the first and at the last BWT position of every LOGICAL run (every endmarker is a run of its own,
// src/r-index.cpp:840-848), so that no suffix array has to outlive the merge that produced it.
struct TextBwt {
    uint64_t n = 0;                                   // symbols of the collection (every sequence ends in \n)
    std::vector<uint64_t> seq_start;                  // text position of every sequence
    std::vector<std::pair<uint8_t, uint64_t>> runs;   // grlBWT-style maximal runs (endmarkers not split)
    std::vector<uint64_t> head, tail;                 // per logical run: text position of its first / last suffix
    uint64_t max_len = 1;
};

// suffix array of one newline-terminated text (row 0 = the extra terminator, row p + 1 = the p-th suffix); endmarkers order by sequence
static void suffix_array_of(const std::vector<uint8_t> &text, std::vector<int32_t> &SA) {
    const uint64_t n = text.size();
    if (n + 1 >= (1ull << 31)) throw Error(PGX_ERR_UNSUPPORTED, "a text longer than 2^31 - 2 symbols (pass the collection as several texts: pgx_build_index_from_texts)");
    uint64_t m = 0;
    for (uint8_t c : text) m += (c == '\n');
    // alphabet: 0 = extra terminator, 1..m = endmarkers by sequence, then present bytes by value
    int32_t rank_of[256];
    bool present[256] = {false};
    for (uint8_t c : text) present[c] = true;
    int32_t K = (int32_t)m + 1;
    for (int c = 0; c < 256; c++)
        if (present[c] && c != '\n') rank_of[c] = K++;
    std::vector<int32_t> s(n + 1);
    SA.resize(n + 1);
    int32_t seq = 0;
    for (uint64_t i = 0; i < n; i++) s[i] = text[i] == '\n' ? ++seq : rank_of[text[i]];
    s[n] = 0;
    Sais::run(s.data(), SA.data(), (int32_t)(n + 1), K);
}

// ---- the collection as several texts ("chromosomes"): one suffix array each, built side by side, then ONE k-way merge of the sorted
// suffix lists, split over threads by sampled splitter suffixes.  A collection of any size builds this way as long as every text stays
// below 2^31 symbols (FastLocate(std::string) takes grlBWT's output of any size, src/r-index.cpp:778-1139); the order is that of one
// suffix array over the concatenation: symbols compare by byte value, an endmarker is smaller than any symbol, two endmarkers
// compare by sequence number.
namespace {
struct Chunk {
    std::vector<uint8_t> text; // newline-terminated, 8 bytes of padding behind it
    std::vector<int32_t> SA;   // n + 1 entries, [0] = the extra terminator
    uint64_t n = 0, base = 0;  // symbols, text position of its first symbol in the collection
    std::vector<uint64_t> seq_start; // local
    uint64_t seq_base = 0;
};
struct Suf { uint32_t c; uint64_t i; };

static inline uint64_t load64(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
static inline uint64_t seq_of(const Chunk &c, uint64_t pos) {
    return c.seq_base + (uint64_t)(std::upper_bound(c.seq_start.begin(), c.seq_start.end(), pos) - c.seq_start.begin()) - 1;
}
// suffix (a, i) < suffix (b, j) in the order of the collection
static bool suffix_less(const Chunk &A, uint64_t i, const Chunk &B, uint64_t j) {
    const uint8_t *a = A.text.data() + i, *b = B.text.data() + j;
    for (uint64_t o = 0;; o += 8) {
        const uint64_t wa = load64(a + o), wb = load64(b + o);
        const uint64_t x = wa ^ wb, v = wa ^ 0x0A0A0A0A0A0A0A0Aull;
        const uint64_t nl = (v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull; // lowest set flag = first \n of wa
        if (!x && !nl) continue;
        const unsigned pd = x ? (unsigned)__builtin_ctzll(x) >> 3 : 8u, pn = nl ? (unsigned)__builtin_ctzll(nl) >> 3 : 8u;
        if (pn < pd) { // equal up to and including an endmarker: the sequences' numbers decide
            const uint64_t sa = seq_of(A, i + o + pn), sb = seq_of(B, j + o + pn);
            return sa < sb;
        }
        return a[o + pd] < b[o + pd]; // (\n = 10 sorts below A C G N T as bytes)
    }
}
// the first 21 symbols of a suffix as one integer that orders like the suffix: 3 bits per symbol, nothing behind an endmarker
struct KeyCodes {
    uint8_t v[256];
    KeyCodes() {
        for (int c = 0; c < 256; c++) v[c] = c < 'A' ? 1 : 6; // (bytes the .ri builder rejects anyway: kept in byte order around the alphabet)
        v[(uint8_t)'\n'] = 0; v[(uint8_t)'A'] = 1; v[(uint8_t)'C'] = 2; v[(uint8_t)'G'] = 3; v[(uint8_t)'N'] = 4; v[(uint8_t)'T'] = 5;
    }
};
static inline uint64_t suffix_key(const Chunk &C, uint64_t i) {
    static const KeyCodes kc;
    const uint8_t *p = C.text.data() + i;
    uint64_t k = 0;
    for (int d = 0; d < 21; d++) {
        const uint64_t v = kc.v[p[d]];
        if (!v) break;
        k |= v << (60 - 3 * d);
    }
    return k;
}
} // namespace

static void merge_chunks(std::vector<Chunk> &ch, TextBwt &o) {
    const size_t C = ch.size();
    uint64_t n = 0, n_seq = 0;
    for (auto &c : ch) { c.base = n; c.seq_base = n_seq; n += c.n; n_seq += c.seq_start.size(); }
    o.n = n;
    o.seq_start.clear();
    for (auto &c : ch) for (uint64_t v : c.seq_start) o.seq_start.push_back(c.base + v);
    unsigned T = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 32u));
    if (const char *e = std::getenv("PGX_BUILD_THREADS")) T = (unsigned)std::max<long>(1, std::min<long>(std::atol(e), 256));
    if (n < (1u << 16)) T = std::min(T, 2u);
    // splitters: T - 1 suffixes that cut the merged order into even parts -- candidates at even ranks of every chunk, sorted, every C-th taken
    std::vector<Suf> cand;
    for (uint32_t c = 0; c < C; c++)
        for (unsigned t = 1; t < T; t++) cand.push_back({c, (uint64_t)ch[c].SA[1 + ch[c].n * t / T]});
    std::sort(cand.begin(), cand.end(), [&](const Suf &a, const Suf &b) { return (a.c != b.c || a.i != b.i) && suffix_less(ch[a.c], a.i, ch[b.c], b.i); });
    std::vector<Suf> split;
    for (unsigned t = 1; t < T; t++) split.push_back(cand[(size_t)t * C - (C + 1) / 2]);
    // lo[t][c] = first rank of chunk c that belongs to part t (suffixes >= splitter t - 1)
    std::vector<std::vector<uint64_t>> lo(T + 1, std::vector<uint64_t>(C, 0));
    for (uint32_t c = 0; c < C; c++) {
        lo[T][c] = ch[c].n;
        for (unsigned t = 1; t < T; t++) {
            const Suf &sp = split[t - 1];
            uint64_t a = 0, b = ch[c].n; // first rank whose suffix is not smaller than the splitter
            while (a < b) {
                const uint64_t mid = (a + b) >> 1, pos = (uint64_t)ch[c].SA[1 + mid];
                const bool less = (c == sp.c && pos == sp.i) ? false : suffix_less(ch[c], pos, ch[sp.c], sp.i);
                if (less) a = mid + 1; else b = mid;
            }
            lo[t][c] = a;
        }
        for (unsigned t = 1; t <= T; t++) lo[t][c] = std::max(lo[t][c], lo[t - 1][c]); // (equal candidates: keep the parts nested)
    }
    struct Part { std::vector<uint8_t> sym; std::vector<uint64_t> len, head, tail; };
    std::vector<Part> parts(T);
    std::vector<std::thread> th;
    std::exception_ptr err;
    std::mutex mu;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            try {
                Part &P = parts[t];
                std::vector<uint64_t> at(C), end(C), key(C), pos(C);
                std::vector<uint8_t> live(C, 0);
                auto load = [&](size_t c) {
                    if (at[c] < end[c]) {
                        pos[c] = (uint64_t)ch[c].SA[1 + at[c]];
                        key[c] = C > 1 ? suffix_key(ch[c], pos[c]) : 0; // (a single text: nothing to compare)
                        live[c] = 1;
                        if (at[c] + 16 < end[c]) __builtin_prefetch(ch[c].text.data() + (uint64_t)ch[c].SA[1 + at[c] + 16]);
                    } else live[c] = 0;
                };
                for (size_t c = 0; c < C; c++) { at[c] = lo[t][c]; end[c] = lo[t + 1][c]; load(c); }
                for (;;) {
                    int best = -1;
                    for (size_t c = 0; c < C; c++) {
                        if (!live[c]) continue;
                        if (best < 0 || key[c] < key[(size_t)best] ||
                            (key[c] == key[(size_t)best] && suffix_less(ch[c], pos[c], ch[(size_t)best], pos[(size_t)best])))
                            best = (int)c;
                    }
                    if (best < 0) break;
                    const Chunk &K = ch[(size_t)best];
                    const uint64_t i = pos[(size_t)best], g = K.base + i;
                    const uint8_t sym = i ? K.text[i - 1] : (uint8_t)'\n'; // the symbol before a text is the endmarker of the sequence before it
                    if (!P.sym.empty() && P.sym.back() == sym && sym != '\n') { P.len.back()++; P.tail.back() = g; }
                    else { P.sym.push_back(sym); P.len.push_back(1); P.head.push_back(g); P.tail.push_back(g); }
                    at[(size_t)best]++;
                    load((size_t)best);
                }
            } catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); }
        });
    for (auto &x : th) x.join();
    if (err) std::rethrow_exception(err);
    // stitch the parts: a logical run that continues over a part border is one run
    uint64_t R = 0;
    for (auto &P : parts) R += P.sym.size();
    o.runs.clear(); o.head.clear(); o.tail.clear();
    o.head.reserve(R); o.tail.reserve(R);
    uint8_t last_sym = 0;
    bool have = false;
    for (auto &P : parts) {
        for (size_t r = 0; r < P.sym.size(); r++) {
            const uint8_t sym = P.sym[r];
            if (have && last_sym == sym && sym != '\n') { // continues the previous logical (and file) run
                o.runs.back().second += P.len[r];
                o.tail.back() = P.tail[r];
            } else {
                if (have && last_sym == sym) o.runs.back().second += P.len[r]; // endmarkers: one file run, a logical run each
                else o.runs.emplace_back(sym, P.len[r]);
                o.head.push_back(P.head[r]);
                o.tail.push_back(P.tail[r]);
            }
            o.max_len = std::max(o.max_len, o.runs.back().second);
            last_sym = sym; have = true;
        }
        Part().sym.swap(P.sym); std::vector<uint64_t>().swap(P.len); std::vector<uint64_t>().swap(P.head); std::vector<uint64_t>().swap(P.tail);
    }
}

static void load_chunk(const char *path, Chunk &c) {
    c.text = read_whole_file(path);
    if (!c.text.empty() && c.text.back() != '\n') c.text.push_back('\n');
    c.n = c.text.size();
    if (c.n == 0) throw Error(PGX_ERR_FORMAT, std::string("empty text: ") + path);
    c.seq_start.clear();
    c.seq_start.push_back(0);
    for (uint64_t i = 0; i + 1 < c.n; i++)
        if (c.text[i] == '\n') c.seq_start.push_back(i + 1);
    c.text.resize(c.n + 32, 0); // the 8-byte loads of suffix_less and the 21-symbol keys stop at the final endmarker, inside the padding at the latest
}

static void build_texts_bwt(const char *const *text_paths, uint32_t n_texts, TextBwt &o) {
    BuildTimer bt;
    std::vector<Chunk> ch(n_texts);
    std::vector<std::thread> th;
    std::exception_ptr err;
    std::mutex mu;
    for (uint32_t c = 0; c < n_texts; c++)
        th.emplace_back([&, c]() {
            try {
                load_chunk(text_paths[c], ch[c]);
                std::vector<uint8_t> bare(ch[c].text.begin(), ch[c].text.begin() + (std::ptrdiff_t)ch[c].n);
                suffix_array_of(bare, ch[c].SA);
            } catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); }
        });
    for (auto &x : th) x.join();
    if (err) std::rethrow_exception(err);
    bt.lap("suffix arrays (per text)");
    merge_chunks(ch, o);
    bt.lap("k-way merge + runs");
}

static void build_text_bwt(const char *text_path, TextBwt &o) {
    const char *one[1] = {text_path};
    build_texts_bwt(one, 1, o);
}
static void write_rlbwt(const char *out_rlbwt_path, const TextBwt &b) {
    uint64_t bl = 1;
    while (bl < 8 && (b.max_len >> (8 * bl))) bl++;
    std::vector<uint8_t> out;
    out.reserve(16 + b.runs.size() * (1 + bl));
    put<uint64_t>(out, 1);
    put<uint64_t>(out, bl);
    for (auto &r : b.runs) {
        out.push_back(r.first);
        for (uint64_t k = 0; k < bl; k++) out.push_back((uint8_t)(r.second >> (8 * k)));
    }
    write_whole_file(out_rlbwt_path, out);
}

extern "C" pgx_status pgx_build_rlbwt(const char *text_path, const char *out_rlbwt_path) {
    PGX_GUARD_BEGIN
    if (!text_path || !out_rlbwt_path) throw Error(PGX_ERR_ARG, "pgx_build_rlbwt: null argument");
    TextBwt b;
    build_text_bwt(text_path, b);
    write_rlbwt(out_rlbwt_path, b);
    return PGX_OK;
    PGX_GUARD_END
}

// ------------------------------------------------------------------------------------------
static std::vector<std::pair<uint8_t, uint64_t>> read_rlbwt(const std::string &path) {
    std::vector<uint8_t> f = read_whole_file(path);
    ByteReader r(f.data(), f.size());
    uint64_t bs = r.get<uint64_t>("rl_bwt header"), bl = r.get<uint64_t>("rl_bwt header");
    if (bs == 0 || bs > 8 || bl == 0 || bl > 8) throw Error(PGX_ERR_FORMAT, "rl_bwt: bad record widths");
    if ((f.size() - 16) % (bs + bl)) throw Error(PGX_ERR_FORMAT, "rl_bwt: size is not a whole number of records");
    std::vector<std::pair<uint8_t, uint64_t>> runs;
    while (r.o < r.n) {
        uint64_t sym = 0, len = 0;
        std::memcpy(&sym, r.p + r.o, bs);
        std::memcpy(&len, r.p + r.o + bs, bl);
        r.o += bs + bl;
        if (sym > 255) throw Error(PGX_ERR_UNSUPPORTED, "rl_bwt: symbol > 255");
        runs.emplace_back((uint8_t)sym, len);
    }
    return runs;
}

// FastLocate(std::string) + serialize[_encoded].  The SA samples come from the reference's own procedure (a psi walk over
// every sequence, src/r-index.cpp:993-1130) or, when the caller still holds the suffix array the BWT was made from (`tb`),
// straight from it: the sample of BWT position p is the (sequence, offset) of suffix SA[p] -- the same values, without
// n binary searches over the run starts.
static void build_rindex_core(const std::vector<std::pair<uint8_t, uint64_t>> &file_runs, const TextBwt *tb, const char *out_ri_path, int encoded) {
    BuildTimer bt;
    // calculate_C, r-index.hpp:440-482: sym_map = rank among present byte values; C = exclusive sums
    uint64_t freq[256] = {0}, n = 0;
    for (auto &r : file_runs) { freq[r.first] += r.second; n += r.second; }
    uint8_t sym_map[256] = {0};
    std::vector<uint64_t> C;
    {
        uint64_t acc = 0;
        uint8_t k = 0;
        for (int c = 0; c < 256; c++)
            if (freq[c]) { sym_map[c] = k++; C.push_back(acc); acc += freq[c]; }
    }
    const uint64_t sigma = C.size();
    for (int c = 0; c < 256; c++)
        if (freq[c] && code_of_byte((uint8_t)c) < 0) throw Error(PGX_ERR_UNSUPPORTED, "BWT symbol outside {\\n,A,C,G,N,T}");
    if (!freq[(uint8_t)'\n']) throw Error(PGX_ERR_UNSUPPORTED, "BWT without endmarkers");
    // logical runs: every endmarker is its own run (src/r-index.cpp:840-848), blocks of 10 (:312)
    struct Blk { std::vector<uint64_t> cum; std::vector<std::pair<uint8_t, uint64_t>> runs; uint64_t start; };
    std::vector<Blk> blocks;
    std::vector<uint64_t> cum(sigma, 0);
    uint64_t pos = 0, total_runs = 0;
    auto push_run = [&](uint8_t sym, uint64_t len) {
        if (blocks.empty() || blocks.back().runs.size() == 10) blocks.push_back(Blk{cum, {}, pos});
        blocks.back().runs.emplace_back(sym, len);
        cum[sym_map[sym]] += len;
        pos += len;
        total_runs++;
    };
    for (auto &r : file_runs) {
        if (r.second == 0) continue;
        if (r.first == '\n') for (uint64_t i = 0; i < r.second; i++) push_run('\n', 1);
        else push_run(r.first, r.second);
    }
    const uint64_t n_file_blocks = total_runs / 10 + 1; // blocks.resize((total_runs / block_size) + 1), :802
    std::vector<uint8_t> out;
    put<uint32_t>(out, 0x6B3741D8u);
    put<uint32_t>(out, 1);
    // ---- SA samples (src/r-index.cpp:993-1130): walk every sequence backwards with psi, record the text
    //      position at the first (head) and last (tail) BWT position of every logical run
    std::vector<uint64_t> run_start, run_cum;
    std::vector<uint8_t> run_sym;
    {
        std::vector<uint64_t> seen(256, 0);
        for (auto &b : blocks)
            for (auto &ru : b.runs) {
                run_start.push_back(run_start.empty() ? 0 : 0); // patched below
                run_sym.push_back(ru.first);
                run_cum.push_back(seen[ru.first]);
                seen[ru.first] += ru.second;
            }
        uint64_t p = 0, k = 0;
        for (auto &b : blocks)
            for (auto &ru : b.runs) { run_start[k++] = p; p += ru.second; }
    }
    auto run_of = [&](uint64_t p) { return (uint64_t)(std::upper_bound(run_start.begin(), run_start.end(), p) - run_start.begin()) - 1; };
    auto run_len = [&](uint64_t r) { return (r + 1 < run_start.size() ? run_start[r + 1] : n) - run_start[r]; };
    auto psi = [&](uint64_t p, uint64_t r) { return C[sym_map[run_sym[r]]] + run_cum[r] + (p - run_start[r]); }; // r-index.cpp:529-532
    const uint64_t n_seq = freq[(uint8_t)'\n']; // tot_strings()
    struct Sample { uint64_t seq_id, seq_offset, run_id; };
    std::vector<Sample> heads, tails;
    uint64_t max_length = 1; // Header(): max_length(1)
    if (tb) {
        // sequence starts in the text; an endmarker belongs to its sequence (offset = length)
        const std::vector<uint64_t> &seq_start = tb->seq_start;
        const uint64_t R = run_start.size();
        if (seq_start.size() != n_seq || tb->n != n || tb->head.size() != R || tb->tail.size() != R) throw Error(PGX_ERR_FORMAT, "suffix samples and BWT runs disagree");
        for (uint64_t i = 0; i < n_seq; i++) max_length = std::max(max_length, (i + 1 < n_seq ? seq_start[i + 1] : n) - seq_start[i]);
        auto sample_of = [&](uint64_t t, uint64_t r) { // text position t of a suffix -> (sequence, offset)
            const uint64_t q = (uint64_t)(std::upper_bound(seq_start.begin(), seq_start.end(), t) - seq_start.begin()) - 1;
            return Sample{q, t - seq_start[q], r};
        };
        heads.resize(R);
        tails.resize(R);
        unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 16u));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                for (uint64_t r = R * t / nt; r < R * (t + 1) / nt; r++) {
                    heads[r] = sample_of(tb->head[r], r);
                    tails[r] = sample_of(tb->tail[r], r);
                }
            });
        for (auto &t : th) t.join();
        // the reference's walk also records a tail at BWT position n_seq - 1 whether or not a run ends there (:1036); the first n_seq BWT
        // positions are the endmarker suffixes in sequence order, so that suffix is the endmarker of the last sequence
        if (n_seq && run_start[run_of(n_seq - 1)] + run_len(run_of(n_seq - 1)) - 1 != n_seq - 1) tails.push_back(sample_of(n - 1, run_of(n_seq - 1)));
    } else {
        // run ids of the first n_seq BWT positions by symbol change (src/r-index.cpp:993-1008)
        std::vector<uint64_t> endmarker_runs(n_seq, 0);
        {
            uint64_t rid = 0;
            uint8_t prev = n_seq ? run_sym[run_of(0)] : 0;
            for (uint64_t i = 1; i < n_seq; i++) {
                const uint8_t cur = run_sym[run_of(i)];
                if (cur == '\n' || cur != prev) { rid++; prev = cur; }
                endmarker_runs[i] = rid;
            }
        }
        std::mutex mu;
        std::atomic<uint64_t> next_seq{0};
        std::atomic<bool> bad{false};
        auto worker = [&]() {
            for (;;) {
                const uint64_t i = next_seq.fetch_add(1);
                if (i >= n_seq) break;
                std::vector<Sample> hb, tb;
                uint64_t seq_offset = 0, rid = endmarker_runs[i];
                if (i == 0 || rid != endmarker_runs[i - 1]) hb.push_back({i, seq_offset, rid});
                if (i + 1 >= n_seq || rid != endmarker_runs[i + 1]) tb.push_back({i, seq_offset, rid});
                uint64_t r = run_of(i);
                uint8_t sym = run_sym[r];
                uint64_t pos = psi(i, r);
                seq_offset++;
                while (sym != '\n') {
                    r = run_of(pos);
                    const uint64_t first = run_start[r], last = first + run_len(r) - 1;
                    if (pos == first) hb.push_back({i, seq_offset, r});
                    if (pos == last) tb.push_back({i, seq_offset, r});
                    sym = run_sym[r];
                    pos = psi(pos, r);
                    seq_offset++;
                    if (seq_offset > n + 1) { bad.store(true); break; } // not a BWT of terminated sequences
                }
                for (auto &x : hb) x.seq_offset = seq_offset - 1 - x.seq_offset;
                for (auto &x : tb) x.seq_offset = seq_offset - 1 - x.seq_offset;
                std::lock_guard<std::mutex> lk(mu);
                max_length = std::max(max_length, seq_offset);
                heads.insert(heads.end(), hb.begin(), hb.end());
                tails.insert(tails.end(), tb.begin(), tb.end());
            }
        };
        unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 16u));
        if (nt > n_seq) nt = (unsigned)std::max<uint64_t>(1, n_seq);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(worker);
        for (auto &t : th) t.join();
        if (bad.load()) throw Error(PGX_ERR_FORMAT, "rl_bwt is not the BWT of a newline-terminated collection (psi walk does not end)");
    }
    bt.lap(".ri: blocks + SA samples");
    if (heads.size() < total_runs || tails.size() < total_runs)
        throw Error(PGX_ERR_FORMAT, "rl_bwt is not the BWT of a newline-terminated collection (sampling walk incomplete)");
    if (!tb) std::stable_sort(heads.begin(), heads.end(), [](const Sample &a, const Sample &b) { return a.run_id < b.run_id; }); // (already in run order otherwise)
    // by text position (sequence, offset): distinct per sample, spread over the whole text
    parallel_range_sort(tails, n_seq * max_length, [&](const Sample &a) { return a.seq_id * max_length + a.seq_offset; });
    auto bits_length = [](uint64_t x) { return (uint8_t)(x ? hi_bit(x) + 1 : 0); }; // sdsl::bits::length
    auto pack = [&](uint64_t id, uint64_t off) { return id * max_length + off; };   // r-index.hpp:420-422
    put<uint64_t>(out, max_length);
    put<uint64_t>(out, encoded ? 1ull : 0ull); // flags
    {
        std::vector<uint64_t> sv(total_runs), l2r(total_runs);
        SdVector last;
        last.size = n_seq * max_length; // sd_vector_builder(n_seq * max_length, total_runs)
        last.ones.resize(total_runs);
        {
            const unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 16u));
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    for (uint64_t i = total_runs * t / nt; i < total_runs * (t + 1) / nt; i++) {
                        sv[i] = pack(heads[i].seq_id, heads[i].seq_offset);
                        last.ones[i] = pack(tails[i].seq_id, tails[i].seq_offset);
                        l2r[i] = tails[i].run_id;
                    }
                });
            for (auto &t : th) t.join();
        }
        uint8_t w = bits_length(pack(n_seq - 1, max_length - 1));
        IntVector::pack(sv, w ? w : 64).write(out, true);
        last.write(out);
        uint8_t w2 = bits_length(total_runs - 1);
        IntVector::pack(l2r, w2 ? w2 : 64).write(out, true);
    }
    {
        std::vector<uint64_t> sm(256);
        for (int c = 0; c < 256; c++) sm[c] = sym_map[c];
        IntVector::pack(sm, 8).write(out, false);
        IntVector::pack(C, 64).write(out, false);
    }
    bt.lap(".ri: sort + pack samples");
    SdVector bsp;
    bsp.size = n;
    for (auto &b : blocks) bsp.ones.push_back(b.start);
    bsp.write(out);
    put<uint64_t>(out, n);
    if (encoded) { // serialize_encoded, src/r-index.cpp:312-372
        put<uint64_t>(out, 10);
        bool hasN = freq[(uint8_t)'N'] != 0;
        put<uint8_t>(out, hasN ? 1 : 0);
        std::vector<uint8_t> stream;
        std::vector<uint64_t> offs;
        for (uint64_t b = 0; b < n_file_blocks; b++) {
            offs.push_back(stream.size());
            if (b < blocks.size()) {
                for (uint64_t i = 0; i < sigma; i++) bytecode_write(stream, blocks[b].cum[i]);
                for (auto &ru : blocks[b].runs) {
                    uint64_t prefix = std::min<uint64_t>(ru.second - 1, 31);
                    stream.push_back((uint8_t)((code_of_byte(ru.first) << 5) | (int)prefix));
                    if (prefix == 31) bytecode_write(stream, ru.second - 32);
                }
            } else {
                for (int i = 0; i < 8; i++) bytecode_write(stream, 0); // Run_blocks(): character_cum_ranks(8), hpp:144
            }
        }
        unsigned w = offs.back() ? hi_bit(offs.back()) + 1 : 64; // sdsl::bits::length; width(0) keeps 64
        IntVector::pack(offs, (uint8_t)w).write(out, true);
        put<uint64_t>(out, stream.size());
        out.insert(out.end(), stream.begin(), stream.end());
    } else { // serialize, src/r-index.cpp:284-291 ; Run_blocks::serialize r-index.hpp:261-277
        put<uint64_t>(out, n_file_blocks);
        for (uint64_t b = 0; b < n_file_blocks; b++) {
            if (b < blocks.size()) {
                IntVector::pack(blocks[b].cum, 64).write(out, false);
                put<uint64_t>(out, blocks[b].runs.size());
                for (auto &ru : blocks[b].runs) { put<uint64_t>(out, ru.first); put<uint64_t>(out, ru.second); }
            } else {
                IntVector::pack(std::vector<uint64_t>(8, 0), 64).write(out, false);
                put<uint64_t>(out, 0);
            }
        }
    }
    bt.lap(".ri: blocks stream");
    write_whole_file(out_ri_path, out);
    bt.lap(".ri: file written");
}

extern "C" pgx_status pgx_build_rindex(const char *rlbwt_path, const char *out_ri_path, int encoded) {
    PGX_GUARD_BEGIN
    if (!rlbwt_path || !out_ri_path) throw Error(PGX_ERR_ARG, "pgx_build_rindex: null argument");
    build_rindex_core(read_rlbwt(rlbwt_path), nullptr, out_ri_path, encoded);
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_build_index_from_text(const char *text_path, const char *out_rlbwt_path, const char *out_ri_path, int encoded) {
    PGX_GUARD_BEGIN
    if (!text_path || !out_ri_path) throw Error(PGX_ERR_ARG, "pgx_build_index_from_text: null argument");
    TextBwt b;
    build_text_bwt(text_path, b);
    if (out_rlbwt_path) write_rlbwt(out_rlbwt_path, b);
    build_rindex_core(b.runs, &b, out_ri_path, encoded);
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_build_index_from_texts(const char *const *text_paths, uint32_t n_texts, const char *out_rlbwt_path, const char *out_ri_path, int encoded) {
    PGX_GUARD_BEGIN
    if (!text_paths || !n_texts || !out_ri_path) throw Error(PGX_ERR_ARG, "pgx_build_index_from_texts: null argument");
    for (uint32_t i = 0; i < n_texts; i++)
        if (!text_paths[i]) throw Error(PGX_ERR_ARG, "pgx_build_index_from_texts: null path");
    TextBwt b;
    build_texts_bwt(text_paths, n_texts, b);
    if (out_rlbwt_path) write_rlbwt(out_rlbwt_path, b);
    build_rindex_core(b.runs, &b, out_ri_path, encoded);
    return PGX_OK;
    PGX_GUARD_END
}

// ------------------------------------------------------------------------------------------
namespace pgx {
// max_node_floor: the item width covers at least this node id (merge_tags takes the graph's largest id, merge_tags.cpp:627-638)
void write_compact_tags(const char *out_path, const uint64_t *values, const uint64_t *lengths, uint64_t n_runs, uint64_t max_node_floor) {
    std::vector<uint64_t> items;
    SdVector starts, intervals;
    uint64_t bwt_pos = 0, max_node = max_node_floor;
    auto emit = [&](uint64_t v, uint64_t len) {
        intervals.ones.push_back(bwt_pos);
        bwt_pos += len;
        if (items.size() % 10 == 0) starts.ones.push_back(items.size());
        items.push_back(v);
    };
    for (uint64_t i = 0; i < n_runs; i++) {
        uint64_t len = lengths[i];
        max_node = std::max(max_node, values[i] >> 11);
        while (len >= 512) { emit(values[i], 511); len -= 511; } // max_tag_len = 1 << length_bits, tag_arrays.cpp:941-957
        if (len > 0) emit(values[i], len);
    }
    const unsigned width = 10 + 1 + (hi_bit(max_node) + 1); // merge_tags.cpp:636-637
    starts.size = starts.ones.empty() ? 1 : starts.ones.back() + 1; // builder(start_pos + 1, ones), tag_arrays.cpp:626
    intervals.size = bwt_pos + 1;                                   // builder(cumulative_run_bwt_position + 1, runs), :635
    std::vector<uint8_t> out;
    IntVector::pack(items, (uint8_t)width).write(out, true);
    starts.write(out);
    intervals.write(out);
    write_whole_file(out_path, out);
}
} // namespace pgx

extern "C" pgx_status pgx_write_compact_tags(const char *out_path, const uint64_t *values, const uint64_t *lengths, uint64_t n_runs) {
    PGX_GUARD_BEGIN
    if (!out_path || (n_runs && (!values || !lengths))) throw Error(PGX_ERR_ARG, "pgx_write_compact_tags: null argument");
    write_compact_tags(out_path, values, lengths, n_runs, 0);
    return PGX_OK;
    PGX_GUARD_END
}

// ------------------------------------------------------------------------------------------
// convert_tags equivalent: "algorithm format" (bare ByteCode stream of
// offset:10 | rev:1 | len:9 | node<<20, written by build_tags, src/tag_arrays.cpp:28-36,104-127) ->
//   compact = 0: ByteCode query format  = compressed_serialize (src/tag_arrays.cpp:656-734) +
//                merge_compressed_files (:558-620); reproduces the reference's fixture
//                bidirectional_test/xy_bidirectional_compressed.tags byte for byte
//   compact = 1: sdsl-compact query format (what find_mems.cpp loads): as pgx_write_compact_tags.
// (HEAD's convert_tags.cpp writes compact values into the ByteCode container, which no HEAD loader/query
//  pair decodes consistently -- SURVEY section 5 format 4; that combination is deliberately not offered.)
extern "C" pgx_status pgx_convert_tags(const char *in_path, const char *out_path, int compact) {
    PGX_GUARD_BEGIN
    if (!in_path || !out_path) throw Error(PGX_ERR_ARG, "pgx_convert_tags: null argument");
    std::vector<uint8_t> in = read_whole_file(in_path);
    std::vector<uint64_t> vals, lens;
    uint64_t loc = 0;
    while (loc < in.size()) {
        const uint64_t d = bytecode_read(in.data(), in.size(), loc, "tag run");
        vals.push_back(d);
        lens.push_back((d >> 11) & 0x1FF); // decode_run, length_bits = 9
    }
    if (compact) {
        std::vector<uint64_t> cv(vals.size());
        for (size_t i = 0; i < vals.size(); i++) cv[i] = (vals[i] & 0x7FF) | ((vals[i] >> 20) << 11);
        return pgx_write_compact_tags(out_path, cv.data(), lens.data(), vals.size());
    }
    std::vector<uint8_t> stream;
    SdVector starts, intervals;
    uint64_t bwt_pos = 0, nruns = 0, last_start = 0;
    auto emit = [&](uint64_t v, uint64_t len) {
        intervals.ones.push_back(bwt_pos);
        bwt_pos += len;
        if (nruns % 10 == 0) { last_start = stream.size(); starts.ones.push_back(last_start); }
        bytecode_write(stream, (v & ~(0x1FFull << 11)) | (len << 11)); // encode_run_length with this chunk's length
        nruns++;
    };
    for (size_t i = 0; i < vals.size(); i++) {
        uint64_t len = lens[i];
        while (len >= 512) { emit(vals[i], 511); len -= 511; }
        if (len > 0) emit(vals[i], len);
    }
    starts.size = last_start + 1;  // sd_vector_builder(start_pos + 1, encoded_start_ones), :565
    intervals.size = bwt_pos + 1;  // sd_vector_builder(cumulative_run_bwt_position + 1, runs), :590
    std::vector<uint8_t> out;
    put<uint64_t>(out, stream.size()); // size header patched in by merge_compressed_files (:613-617)
    out.insert(out.end(), stream.begin(), stream.end());
    starts.write(out);
    intervals.write(out);
    write_whole_file(out_path, out);
    return PGX_OK;
    PGX_GUARD_END
}

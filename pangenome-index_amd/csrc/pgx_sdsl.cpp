// pgx_sdsl.cpp -- readers and writers for the serialised SDSL containers the reference's files
// are made of (vgteam/sdsl-lite classic layout; SURVEY section 5 "On-disk formats").  Written from
// the published layout; pinned byte-for-byte by re-serialising the containers of the reference's
// own fixtures (tests/test_formats.py).
#include <algorithm>
#include <cstdio>

#include "pgx_host.hpp"

namespace pgx {

static inline unsigned hi_bit(uint64_t x) { return x ? 63u - (unsigned)__builtin_clzll(x) : 0u; } // sdsl::bits::hi

std::vector<uint8_t> read_whole_file(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw Error(PGX_ERR_IO, "Cannot open file: " + path);
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b((size_t)sz);
    if (sz > 0 && std::fread(b.data(), 1, (size_t)sz, f) != (size_t)sz) {
        std::fclose(f);
        throw Error(PGX_ERR_IO, "Short read: " + path);
    }
    std::fclose(f);
    return b;
}

void write_whole_file(const std::string &path, const std::vector<uint8_t> &bytes) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error(PGX_ERR_IO, "Cannot create file: " + path);
    if (!bytes.empty() && std::fwrite(bytes.data(), 1, bytes.size(), f) != bytes.size()) {
        std::fclose(f);
        throw Error(PGX_ERR_IO, "Short write: " + path);
    }
    std::fclose(f);
}

template <class T> static void put(std::vector<uint8_t> &out, T v) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&v);
    out.insert(out.end(), p, p + sizeof(T));
}

// gbwt::ByteCode (call sites src/r-index.cpp:70,148,325,339; src/tag_arrays.cpp:819,826)
uint64_t bytecode_read(const uint8_t *s, uint64_t n, uint64_t &i, const char *what) {
    uint64_t off = 0, res = 0;
    for (;;) {
        if (i >= n || off > 63) throw Error(PGX_ERR_FORMAT, std::string("bad ByteCode value in ") + what);
        uint8_t b = s[i++];
        res += (uint64_t)(b & 0x7F) << off;
        if (!(b & 0x80)) return res;
        off += 7;
    }
}

void bytecode_write(std::vector<uint8_t> &out, uint64_t v) {
    while (v > 0x7F) {
        out.push_back((uint8_t)((v & 0x7F) | 0x80));
        v >>= 7;
    }
    out.push_back((uint8_t)v);
}

// ---- int_vector ---------------------------------------------------------------------------
void IntVector::read(ByteReader &r, int fixed_width, const char *what) {
    bits = r.get<uint64_t>(what);
    width = fixed_width ? (uint8_t)fixed_width : r.get<uint8_t>(what);
    if (width > 64) throw Error(PGX_ERR_FORMAT, std::string("int_vector width > 64 in ") + what);
    uint64_t nw = (bits + 63) / 64;
    if (nw > (r.n - r.o) / 8 + 1) throw Error(PGX_ERR_FORMAT, std::string("truncated file while reading ") + what);
    r.need(nw * 8, what);
    words.assign(nw + 1, 0);
    if (nw) std::memcpy(words.data(), r.p + r.o, nw * 8);
    r.o += nw * 8;
}

void IntVector::write(std::vector<uint8_t> &out, bool with_width_byte) const {
    put<uint64_t>(out, bits);
    if (with_width_byte) put<uint8_t>(out, width);
    uint64_t nw = (bits + 63) / 64;
    for (uint64_t i = 0; i < nw; i++) put<uint64_t>(out, words[i]);
}

IntVector IntVector::pack(const std::vector<uint64_t> &vals, uint8_t width) {
    IntVector v;
    v.width = width;
    v.bits = (uint64_t)vals.size() * width;
    v.words.assign((v.bits + 63) / 64 + 1, 0);
    for (uint64_t i = 0; i < vals.size(); i++) {
        uint64_t x = width == 64 ? vals[i] : (vals[i] & ((1ULL << width) - 1));
        uint64_t bit = i * width, wd = bit >> 6, sh = bit & 63;
        v.words[wd] |= x << sh;
        if (sh + width > 64) v.words[wd + 1] |= x >> (64 - sh);
    }
    return v;
}

// ---- select_support_mcl -------------------------------------------------------------------
static void skip_select_support(ByteReader &r, const char *what) {
    uint64_t cnt = r.get<uint64_t>(what);
    if (!cnt) return;
    IntVector t;
    t.read(r, 0, what); // superblock
    t.read(r, 1, what); // mini_or_long
    uint64_t sb = (cnt + 4095) >> 12;
    for (uint64_t i = 0; i < sb; i++) t.read(r, 0, what);
}

// select_support_mcl<b,1>::init_slow + serialize over the bit vector `bv` of `nbits` bits.
static void write_select_support(std::vector<uint8_t> &out, const std::vector<uint64_t> &bv, uint64_t nbits, int b) {
    const uint64_t SB = 4096;
    auto bit = [&](uint64_t i) { return (int)((bv[i >> 6] >> (i & 63)) & 1); };
    uint64_t arg_cnt = 0;
    for (uint64_t i = 0; i < nbits; i++) arg_cnt += (uint64_t)(bit(i) == b);
    put<uint64_t>(out, arg_cnt);
    if (!arg_cnt) return;
    uint64_t capacity = ((nbits + 63) >> 6) << 6;
    uint64_t logn = hi_bit(capacity) + 1, logn2 = logn * logn, logn4 = logn2 * logn2;
    uint64_t sb = (arg_cnt + SB - 1) / SB;
    std::vector<uint64_t> superblock(sb, 0);
    std::vector<IntVector> mini(sb), lng(sb);
    std::vector<uint8_t> is_long(sb, 0);
    bool any_long = false;
    std::vector<uint64_t> argpos(SB);
    uint64_t cnt = 0, sb_cnt = 0;
    for (uint64_t i = 0; i < nbits; i++) {
        if (bit(i) != b) continue;
        argpos[cnt % SB] = i;
        ++cnt;
        if (cnt % SB == 0 || cnt == arg_cnt) {
            superblock[sb_cnt] = argpos[0];
            uint64_t last = (cnt - 1) % SB;
            uint64_t pos_diff = argpos[last] - argpos[0];
            if (pos_diff > logn4) {
                any_long = true;
                is_long[sb_cnt] = 1;
                std::vector<uint64_t> vals(SB, 0);
                for (uint64_t j = 0; j <= last; j++) vals[j] = argpos[j];
                lng[sb_cnt] = IntVector::pack(vals, (uint8_t)(hi_bit(argpos[last]) + 1));
            } else {
                std::vector<uint64_t> vals(64, 0);
                for (uint64_t j = 0; j <= last; j += 64) vals[j / 64] = argpos[j] - argpos[0];
                mini[sb_cnt] = IntVector::pack(vals, (uint8_t)(hi_bit(pos_diff) + 1));
            }
            ++sb_cnt;
        }
    }
    IntVector::pack(superblock, (uint8_t)logn).write(out, true);
    if (any_long) {
        std::vector<uint64_t> ml(sb);
        for (uint64_t i = 0; i < sb; i++) ml[i] = !is_long[i];
        IntVector::pack(ml, 1).write(out, false);
    } else {
        put<uint64_t>(out, 0); // empty bit_vector
    }
    for (uint64_t i = 0; i < sb; i++) (is_long[i] ? lng[i] : mini[i]).write(out, true);
}

// ---- sd_vector -----------------------------------------------------------------------------
void SdVector::read(ByteReader &r, const char *what) {
    size = r.get<uint64_t>(what);
    uint8_t wl = r.get<uint8_t>(what);
    if (wl > 63) throw Error(PGX_ERR_FORMAT, std::string("sd_vector low width > 63 in ") + what);
    IntVector low, high;
    low.read(r, 0, what);
    high.read(r, 1, what);
    skip_select_support(r, what);
    skip_select_support(r, what);
    ones.clear();
    uint64_t i = 0;
    for (uint64_t pos = 0; pos < high.bits; pos++) {
        if ((high.words[pos >> 6] >> (pos & 63)) & 1) {
            if (i >= low.size() && wl) throw Error(PGX_ERR_FORMAT, std::string("sd_vector low/high mismatch in ") + what);
            uint64_t hp = pos - i;
            uint64_t lv = wl ? low.get(i) : 0;
            ones.push_back((hp << wl) | lv);
            i++;
        }
    }
    for (uint64_t k = 1; k < ones.size(); k++)
        if (ones[k] <= ones[k - 1]) throw Error(PGX_ERR_FORMAT, std::string("sd_vector not increasing in ") + what);
    if (!ones.empty() && ones.back() >= size && size) throw Error(PGX_ERR_FORMAT, std::string("sd_vector element beyond size in ") + what);
}

// sd_vector_builder(n, m) parameters + sd_vector::serialize
void SdVector::write(std::vector<uint8_t> &out) const {
    uint64_t m = ones.size();
    uint64_t logm = hi_bit(m) + 1, logn = hi_bit(size) + 1;
    if (logm == logn) logm--;
    uint8_t wl = (uint8_t)(logn - logm);
    put<uint64_t>(out, size);
    put<uint8_t>(out, wl);
    std::vector<uint64_t> lowv(m);
    // vgteam/sdsl-lite sizes the high part as m + ceil(n / 2^wl) bits (pinned by the fixtures of xy.ri:
    // blocks_start_pos: 162 ones, n = 8022, wl = 5 -> 413 bits; last: 1620 ones, n = 8024, wl = 2 -> 3626)
    uint64_t high_bits = m + ((size + (1ULL << wl) - 1) >> wl);
    std::vector<uint64_t> high((high_bits + 63) / 64 + 1, 0);
    for (uint64_t i = 0; i < m; i++) {
        lowv[i] = ones[i];
        uint64_t pos = (ones[i] >> wl) + i;
        high[pos >> 6] |= 1ULL << (pos & 63);
    }
    IntVector::pack(lowv, wl).write(out, true);
    IntVector hv;
    hv.width = 1;
    hv.bits = high_bits;
    hv.words = high;
    hv.write(out, false);
    write_select_support(out, high, high_bits, 1);
    write_select_support(out, high, high_bits, 0);
}

} // namespace pgx

// pgx_runtime.hip -- device memory, launches and the batch pipeline behind the C ABI.
//
// Pipeline of pgx_batch_run (one HIP stream, results stay on the device):
//   scan(cap)      -> slot offsets (worst-case MEMs per read: min(len, len - min_len + 1)); the slot buffer
//                     is bounded, larger batches run in chunks of consecutive reads
//   find_mems      -> MEM slots + per-read counts                      [dominant kernel, persistent grid]
//   scan(count)    -> CSR offsets ; compact slots -> dense MEM array in read order
//   tag_locate     -> per MEM run_nums + first item + size-class lists ; scans -> segment offsets
//   tag_small      -> <= 16 runs: gather + sort + unique in registers
//   tag_gather / tag_sort_unique / tag_sort_large -> listed bigger queries (identical large ones once)
//   scan ; tag_compact -> positions CSR
// The only host synchronisations are the scalar read-backs that size the next buffer.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <array>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "pgx_device.h"
#include "pgx_host.hpp"
#include "pgx_runtime.hpp"

// roctx ranges and stage marks for rocprofv3 --marker-trace (the reference's TIME stopwatches, src/find_mems.cpp:20-24,100-136).
// Off unless PGX_ROCTX=1: the library is looked up at run time so that nothing links against the profiler.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    void (*mark)(const char *) = nullptr;
    bool on = false;
};
static const Roctx &roctx() {
    static const Roctx r = [] {
        Roctx x;
        const char *e = std::getenv("PGX_ROCTX");
        if (!e || !std::atoi(e)) return x;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return x;
        x.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        x.mark = reinterpret_cast<void (*)(const char *)>(dlsym(h, "roctxMarkA"));
        x.on = x.push && x.pop && x.mark;
        return x;
    }();
    return r;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(roctx().on) { if (on) roctx().push(name); }
    ~RoctxRange() { if (on) roctx().pop(); }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};


using namespace pgx;

static int checked_device_count() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        throw Error(PGX_ERR_NO_DEVICE, "no usable HIP device (libpgx has no CPU fallback)");
    }
    return n;
}

static void use_device(int device) {
    int n = checked_device_count();
    if (device < 0 || device >= n) throw Error(PGX_ERR_ARG, "device ordinal out of range");
    HIPCHECK(hipSetDevice(device));
}

struct pgx_device_image {
    int device = -1;
    PgxDevImage img{};
    DevBuf blocks, dir, blow, consts, tstart, tvals, tdir, tpair, tbucket, seed, seed_small, seed_end, exc, pairs, first_ext, sbase2, pbase;
    DevBuf rstart, rsamp, rdir, lpos, lnext, ldir; // locate image, uploaded on first use
    DevBuf lce_sa, lce_text, lce_flags, lce_lcp;   // LCE image (ensure_lce), built on the first batch
    int lce_state = 0;                             // 0 not tried, 1 built, 2 not available for this index / device
    DevBuf lit_bstart, lit_cum, lit_runs, lit_roff, lit_tabs; // literal count image (quirk 3), uploaded on first use
    PgxLitImage lit{};
    bool has_lit = false;
    PgxLocImage loc{};
    bool has_loc = false;
    size_t lds_bytes = 0; // dynamic LDS of the LDS-image kernels (0 = image stays in global memory)
};

void pgx_release_device_images(pgx_index *h) {
    for (auto *d : h->dev) {
        if (!d) continue;
        if (hipSetDevice(d->device) == hipSuccess) {
            d->blocks.release(); d->dir.release(); d->blow.release(); d->consts.release();
            d->tstart.release(); d->tvals.release(); d->tdir.release(); d->tpair.release(); d->tbucket.release(); d->seed.release(); d->seed_small.release(); d->seed_end.release(); d->exc.release(); d->pairs.release(); d->first_ext.release(); d->sbase2.release(); d->pbase.release();
            d->lit_bstart.release(); d->lit_cum.release(); d->lit_runs.release(); d->lit_roff.release(); d->lit_tabs.release();
            d->rstart.release(); d->rsamp.release(); d->rdir.release(); d->lpos.release(); d->lnext.release(); d->ldir.release();
            d->lce_sa.release(); d->lce_text.release(); d->lce_flags.release(); d->lce_lcp.release();
        }
        delete d;
    }
    h->dev.clear();
}

static void upload(DevBuf &b, const void *src, size_t bytes) {
    b.ensure(bytes ? bytes : 16);
    if (bytes) HIPCHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
}

// k-mer seed table of a dense image (pgx_kernels.hip "k-mer seeds"): built level by level on the device,
// 4^L entries at level L, each one pgx_extend of its parent.  K = floor(log4 n), at most 14 (4 GiB of table; chr22 scale, 10 M
// reads, K = 11 / 12 / 13 / 14: 41.2 / 39.1 / 37.2 / 36.4 ms with the 64-byte dense image; n = 64 M, 1 M reads, K = 0 / 9 / 11 / 12: 3.64 / 3.44 /
// 3.14 / 3.07 ms), PGX_SEED_K overrides (0 = no table).
static void build_seed_table(pgx_device_image *d) {
    PgxDevImage &g = d->img;
    // depth: one more than the first at which a random window is expected in the index less than once (4^K >= n), at most 15 (16 GiB): a seed that dies
    // inside the table ends a stage without another trip (n = 640 M: K = 14 / 15 / 16: 20.8 / 19.4-20.5 / 20.1 ms; n = 64 M: K = 12 / 13 / 14: 2.47 / 2.43 / 2.37 ms)
    // Round 4: at most 16 (64 GiB) and three tenths of the device's memory -- with the forward stages through the text a read is ~28 lane trips and the
    // two-step trips behind the seed are a third of them: depth 16 leaves 4 symbols = 2 trips of a 20-symbol step 1 instead of 5 = 3
    // (n = 640 M, K = 15 / 16: main kernel 10.26 / 9.44 ms, 791 / 830 M reads/s, 3.6 s more to build)
    int K = 0;
    while (K < 16 && (K == 0 || (1ull << (2 * (K - 1))) < g.n)) K++;
    {
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) { (void)hipGetLastError(); mem_free = mem_total = (size_t)16 << 30; }
        while (K > 2 && (((size_t)1 << (2 * K)) * sizeof(uint4)) * 5 / 4 > std::min(mem_total * 3 / 10, mem_free / 2)) K--; // table + the level below it while building
    }
    // an image small enough for LDS leaves the loop bound by instruction issue, and every extension a seed replaces is a gain: depth 10
    // (16 MiB of table, hot in L2) whatever n is (x index, 1 M reads, min_len 10, K = 0 / 4 / 6 / 8 / 10: 1.22 / 1.03 / 0.81 / 0.72 / 0.59 ms)
    if (d->lds_bytes) K = 10;
    if (const char *e = std::getenv("PGX_SEED_K")) K = std::atoi(e);
    if (K > PGX_SEED_MAX_K) K = PGX_SEED_MAX_K;
    if (K < 2) return;
    const uint64_t limit = g.n < (1ull << 30) ? (1ull << 32) : (1ull << 40); // what an entry (and the 32-bit kernels) can hold
    DevBuf tmp;
    // end table (stages that start at j = len, i.e. with the extension by 0): depth 8, 1 MiB -- such a stage almost always dies within a few
    // extensions (a read rarely ends where a sequence ends), which the entry's death depth answers at once
    int Ke = std::min(K, 8);
    if (const char *e = std::getenv("PGX_SEED_END_K")) Ke = std::max(0, std::min(std::atoi(e), 12));
    auto build = [&](DevBuf &out, int depth, int end_table) {
        out.ensure(((size_t)1 << (2 * depth)) * sizeof(uint4));
        tmp.ensure(((size_t)1 << (2 * (depth - 1))) * sizeof(uint4));
        for (int L = 0; L < depth; L++) { // level L -> L + 1; level `depth` ends in out
            uint4 *dst = ((depth - (L + 1)) % 2 == 0) ? out.as<uint4>() : tmp.as<uint4>();
            const uint4 *src = ((depth - L) % 2 == 0) ? out.as<uint4>() : tmp.as<uint4>();
            const uint64_t n_dst = 1ull << (2 * (L + 1));
            hipLaunchKernelGGL(pgx_seed_build_kernel, dim3((unsigned)std::min<uint64_t>((n_dst + 255) / 256, 1u << 22)), dim3(256), 0, nullptr, g, src, dst, (uint32_t)L, n_dst, limit, end_table);
            HIPCHECK(hipGetLastError());
        }
        HIPCHECK(hipDeviceSynchronize());
    };
    const int Ks = K > PGX_SEED_SMALL_K ? PGX_SEED_SMALL_K : 0; // second, shallower table for searches with min_len < K
    try {
        build(d->seed, K, 0);
        if (Ks) build(d->seed_small, Ks, 0);
        if (Ke >= 2) build(d->seed_end, Ke, 1);
    } catch (...) { tmp.release(); d->seed.release(); d->seed_small.release(); d->seed_end.release(); throw; }
    tmp.release();
    if (Ke >= 2) { g.seed_end = d->seed_end.as<uint4>(); g.seed_end_k = (uint32_t)Ke; }
    g.seed = d->seed.as<uint4>();
    g.seed_k = (uint32_t)K;
    g.seed_main = g.seed; g.seed_k_main = g.seed_k;
    if (Ks) { g.seed_small = d->seed_small.as<uint4>(); g.seed_k_small = (uint32_t)Ks; }
}

// one device image per (index, device), created on first use; concurrent first calls from several host threads are serialised
static std::mutex g_image_mutex;

static pgx_device_image *device_image(pgx_index *h, int device) {
    use_device(device);
    std::lock_guard<std::mutex> lock(g_image_mutex);
    if ((int)h->dev.size() <= device) h->dev.resize(device + 1, nullptr);
    if (h->dev[device]) return h->dev[device];
    std::unique_ptr<pgx_device_image> d(new pgx_device_image());
    d->device = device;
    const HostImage &m = h->img;
    upload(d->blocks, m.blocks.data(), m.blocks.size());
    upload(d->dir, m.dir.data(), m.dir.size() * 8);
    upload(d->blow, m.blow.data(), m.blow.size() * 2);
    upload(d->exc, m.exc.data(), m.exc.size() * 4);
    upload(d->consts, &m.consts, sizeof(PgxConsts));
    upload(d->tstart, m.tstart.data(), m.tstart.size() * 8);
    upload(d->tvals, m.tvals.data(), m.tvals.size() * 8);
    upload(d->tdir, m.tdir.data(), m.tdir.size() * 4);
    PgxDevImage &g = d->img;
    g.blocks = d->blocks.as<uint4>();
    g.dir = d->dir.as<uint64_t>();
    g.blow = d->blow.as<uint16_t>();
    g.consts = d->consts.as<PgxConsts>();
    g.tstart = d->tstart.as<uint64_t>();
    g.tvals = d->tvals.as<uint64_t>();
    g.tdir = d->tdir.as<uint32_t>();
    g.tpair = nullptr;
    if (!m.tstart.empty() && !m.tvals.empty() && !std::getenv("PGX_NO_TPAIR")) { // tag runs as (start, value) pairs for the locate kernel (built on the device)
        const uint64_t np = std::max<uint64_t>(m.tstart.size(), m.tvals.size());
        d->tpair.ensure(np * sizeof(ulonglong2));
        hipLaunchKernelGGL(pgx_tag_pair_kernel, dim3((unsigned)std::min<uint64_t>((np + 255) / 256, 65536)), dim3(256), 0, nullptr, d->tstart.as<uint64_t>(),
                           d->tvals.as<uint64_t>(), (uint64_t)m.tstart.size(), (uint64_t)m.tvals.size(), d->tpair.as<ulonglong2>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipDeviceSynchronize());
        g.tpair = d->tpair.as<ulonglong2>();
    }
    g.tbucket = nullptr; g.n_tbuckets = 0; g.tbucket_shift = 0;
    if (!m.tstart.empty() && !m.tvals.empty() && !std::getenv("PGX_NO_TBUCKET")) { // tag runs by bucket, one line each (pgx_tag_bucket_kernel): about four runs per bucket
        const uint64_t nr = m.tstart.size(), span = m.tstart.back() + 1;
        uint32_t sh = 0;
        while (sh < 16 && (span >> (sh + 1)) >= nr / 4 + 1) sh++;
        const uint64_t nbk = (span >> sh) + 1;
        if (nbk * 128 <= (16ull << 30)) {
            d->tbucket.ensure(nbk * 128);
            hipLaunchKernelGGL(pgx_tag_bucket_kernel, dim3((unsigned)std::min<uint64_t>((nbk + 255) / 256, 1u << 20)), dim3(256), 0, nullptr, d->tstart.as<uint64_t>(),
                               d->tvals.as<uint64_t>(), nr, (uint64_t)m.tvals.size(), sh, nbk, d->tbucket.as<uint4>());
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipDeviceSynchronize());
            g.tbucket = d->tbucket.as<uint4>(); g.n_tbuckets = nbk; g.tbucket_shift = sh;
        }
    }
    g.n = m.consts.n;
    g.dir_entries = m.consts.dir_entries;
    g.n_tag_runs = m.consts.n_tag_runs;
    g.n_tag_items = m.tvals.size();
    g.tag_dir_entries = m.consts.tag_dir_entries;
    g.n_blocks = m.consts.n_blocks;
    g.dir_shift = m.consts.dir_shift;
    g.excl_mask = m.consts.excl_mask;
    g.tag_dir_shift = m.consts.tag_dir_shift;
    g.dense = m.consts.image_kind; // PGX_IMAGE_RL / _DENSE / _DENSE2
    g.wide = m.consts.wide;
    if (g.dense == PGX_IMAGE_DENSE2 && g.wide) g.dense = 3; // dense2 blocks with delta counts: the 64-bit kernels (pgx_image.h "WIDE")
    upload(d->sbase2, m.sbase2.data(), m.sbase2.size() * 8);
    g.sbase2 = d->sbase2.as<uint64_t>();
    g.d2_sb_shift = m.consts.d2_sb_shift; g.n_sb2 = m.consts.n_sb2;
    g.pbase = nullptr; g.pairs_sb_shift = 0; g.n_sbp = 0; g.pairs_stride = 0;
    g.exc = d->exc.as<uint32_t>();
    size_t img_bytes = m.blocks.size() + m.dir.size() * 8 + m.blow.size() * 2;
    if (g.dense == 1) img_bytes = (size_t)m.consts.n_blocks * 16 * PGX_DENSE_LDS_U4 + 16; // padded blocks, no directory (pgx_dense_load)
    d->lds_bytes = (g.dense < 2 && img_bytes <= 48 * 1024) ? ((img_bytes + 15) & ~(size_t)15) : 0; // the dense2 image is never staged in LDS
    g.seed_k = 0;
    g.seed = nullptr;
    g.seed_end_k = 0;
    g.seed_end = nullptr;
    g.seed_k_main = g.seed_k_small = 0;
    g.seed_main = g.seed_small = nullptr;
    g.pairs = nullptr; g.first_ext = nullptr; g.pair_runs = 0;
    g.lce_sa = nullptr; g.lce_text = nullptr; g.lce_flags = nullptr; g.lce_lcp = nullptr; g.lce_max = 0; g.refill_min = 1;
    if (g.dense && h->has_rank) build_seed_table(d.get());
    if (m.consts.has_pairs && !m.pairs.empty() && h->has_rank) { // the two-step image next to dense2 (pgx_image.h)
        upload(d->pairs, m.pairs.data(), m.pairs.size());
        upload(d->pbase, m.pbase.data(), m.pbase.size() * 8);
        g.pbase = d->pbase.as<uint64_t>();
        g.pairs_sb_shift = m.consts.pairs_sb_shift; g.n_sbp = m.consts.n_sbp;
        g.pairs_stride = m.consts.pairs_stride;
        d->first_ext.ensure(512 * sizeof(uint4));
        hipLaunchKernelGGL(pgx_first_ext_kernel, dim3(1), dim3(256), 0, nullptr, g, d->first_ext.as<uint4>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipDeviceSynchronize());
        g.pairs = d->pairs.as<uint4>();
        g.first_ext = d->first_ext.as<uint4>();
        g.pair_runs = m.consts.pair_runs;
    }
    h->dev[device] = d.release();
    return h->dev[device];
}

static bool literal_count(const pgx_index *h);
static pgx_device_image *literal_image(pgx_index *h, int device);

extern "C" pgx_status pgx_index_to_device(pgx_index *h, int device) {
    PGX_GUARD_BEGIN
    if (!h) throw Error(PGX_ERR_ARG, "pgx_index_to_device: null index");
    (void)device_image(h, device);
    return PGX_OK;
    PGX_GUARD_END
}

static void ensure_lce(pgx_index *h, pgx_device_image *d);
extern "C" pgx_status pgx_index_device_view(pgx_index *h, int device, int which, void *out, uint64_t bytes) {
    PGX_GUARD_BEGIN
    if (!h || !out) throw Error(PGX_ERR_ARG, "pgx_index_device_view: null argument");
    pgx_device_image *d = device_image(h, device);
    const HostImage &m = h->img;
    const void *src = nullptr;
    uint64_t have = 0;
    switch (which) {
    case 0: src = d->blocks.p; have = m.blocks.size(); break;
    case 15: src = d->exc.p; have = m.exc.size() * 4; break;
    case 20: src = d->pairs.p; have = m.pairs.size(); break;
    case 22: src = d->sbase2.p; have = m.sbase2.size() * 8; break;
    case 23: src = d->pbase.p; have = m.pbase.size() * 8; break;
    case 30: case 31: case 32: case 33: { // the LCE image (device only; nothing where it does not exist for this index)
        if (h->has_rank) ensure_lce(h, d);
        if (d->img.lce_sa) {
            const uint64_t n = d->img.n, n_words = (n + 15) / 16 + 64;
            if (which == 30) { src = d->lce_sa.p; have = n * 4; }
            else if (which == 31) { src = d->lce_text.p; have = n_words * 4; }
            else if (which == 32) { src = d->lce_flags.p; have = (n_words / 1024 + 2) * 4; }
            else if (d->img.lce_lcp) { src = d->lce_lcp.p; have = n; }
        }
        break;
    }
    default: throw Error(PGX_ERR_ARG, "pgx_index_device_view: unknown view");
    }
    const uint64_t k = std::min(bytes, have);
    if (k && src) HIPCHECK(hipMemcpy(out, src, k, hipMemcpyDeviceToHost));
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_host_alloc(size_t bytes, void **out) {
    PGX_GUARD_BEGIN
    if (!out) throw Error(PGX_ERR_ARG, "pgx_host_alloc: null argument");
    *out = nullptr;
    (void)checked_device_count();
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); throw Error(PGX_ERR_NOMEM, std::string("hipHostMalloc failed: ") + hipGetErrorString(e)); }
    *out = p;
    return PGX_OK;
    PGX_GUARD_END
}
extern "C" void pgx_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

extern "C" pgx_status pgx_device_count(int *n) {
    PGX_GUARD_BEGIN
    if (!n) throw Error(PGX_ERR_ARG, "pgx_device_count: null argument");
    *n = 0;
    *n = checked_device_count();
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_device_name(int device, char *buf, size_t buflen) {
    PGX_GUARD_BEGIN
    if (!buf || !buflen) throw Error(PGX_ERR_ARG, "pgx_device_name: null argument");
    use_device(device);
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, device));
    std::snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return PGX_OK;
    PGX_GUARD_END
}

// ------------------------------------------------------------------------------------------
// exclusive scan helper: out[n+1] on device (out[n] = total); returns nothing, async on `s`
static void scan_excl(int mode, const void *in, uint64_t n, uint64_t min_len, uint64_t *out, DevBuf &tmp, hipStream_t s,
                      uint64_t *total_out = nullptr, const uint64_t *n_dev = nullptr) {
    if (n == 0) {
        HIPCHECK(hipMemsetAsync(out, 0, 8, s));
        if (total_out) HIPCHECK(hipMemsetAsync(total_out, 0, 8, s));
        return;
    }
    // one launch (decoupled look-back, pgx_scan_onepass_kernel); the tile words carry an epoch, so the buffer is cleared only when it is new or
    // the 18-bit epoch comes round.  PGX_SCAN_THREE=1: the three-launch form (partial sums, their scan, apply)
    static const bool three = [] { const char *e = std::getenv("PGX_SCAN_THREE"); return e && e[0] == '1'; }();
    if (!three) {
        const uint64_t nt = (n + PGX_SCAN1_TILE_ITEMS - 1) / PGX_SCAN1_TILE_ITEMS;
        tmp.ensure((nt + 2) * 8);
        if (tmp.scan_epoch == 0 || tmp.scan_epoch >= 0x3FFFFu) {
            HIPCHECK(hipMemsetAsync(tmp.p, 0, tmp.cap, s));
            tmp.scan_epoch = 0;
        }
        tmp.scan_epoch++;
        unsigned long long *st = tmp.as<unsigned long long>();
        const uint32_t ep = tmp.scan_epoch;
        switch (mode) {
        case 0: hipLaunchKernelGGL(pgx_scan_onepass_kernel<0>, dim3((unsigned)nt), dim3(256), 0, s, in, n, min_len, out, total_out, n_dev, st, ep); break;
        case 1: hipLaunchKernelGGL(pgx_scan_onepass_kernel<1>, dim3((unsigned)nt), dim3(256), 0, s, in, n, min_len, out, total_out, n_dev, st, ep); break;
        case 2: hipLaunchKernelGGL(pgx_scan_onepass_kernel<2>, dim3((unsigned)nt), dim3(256), 0, s, in, n, min_len, out, total_out, n_dev, st, ep); break;
        case 3: hipLaunchKernelGGL(pgx_scan_onepass_kernel<3>, dim3((unsigned)nt), dim3(256), 0, s, in, n, min_len, out, total_out, n_dev, st, ep); break;
        case 4: hipLaunchKernelGGL(pgx_scan_onepass_kernel<4>, dim3((unsigned)nt), dim3(256), 0, s, in, n, min_len, out, total_out, n_dev, st, ep); break;
        default: hipLaunchKernelGGL(pgx_scan_onepass_kernel<5>, dim3((unsigned)nt), dim3(256), 0, s, in, n, min_len, out, total_out, n_dev, st, ep); break;
        }
        HIPCHECK(hipGetLastError());
        return;
    }
    const uint64_t nb = (n + PGX_SCAN_BLOCK_ITEMS - 1) / PGX_SCAN_BLOCK_ITEMS;
    tmp.ensure((nb + 1) * 8);
    tmp.scan_epoch = 0; // (the tile words are overwritten)
    uint64_t *sums = tmp.as<uint64_t>();
    hipLaunchKernelGGL(pgx_scan_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, mode, in, n, min_len, sums, n_dev);
    const int raw = nb <= 2048; // up to 4 M items: no separate scan of the block totals
    if (!raw) hipLaunchKernelGGL(pgx_scan_sums_kernel, dim3(1), dim3(256), 0, s, sums, nb);
    hipLaunchKernelGGL(pgx_scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, s, mode, in, n, min_len, (const uint64_t *)sums, nb, out, total_out, raw, n_dev);
    HIPCHECK(hipGetLastError());
}

void pgx_use_device(int device) { use_device(device); }
void pgx_scan_u64(const uint64_t *in, uint64_t n, uint64_t *out, uint64_t *tmp, hipStream_t s) {
    if (n == 0) { HIPCHECK(hipMemsetAsync(out, 0, 8, s)); return; }
    const uint64_t nb = (n + PGX_SCAN_BLOCK_ITEMS - 1) / PGX_SCAN_BLOCK_ITEMS;
    hipLaunchKernelGGL(pgx_scan_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, 1, (const void *)in, n, (uint64_t)0, tmp, (const uint64_t *)nullptr);
    const int raw = nb <= 2048;
    if (!raw) hipLaunchKernelGGL(pgx_scan_sums_kernel, dim3(1), dim3(256), 0, s, tmp, nb);
    hipLaunchKernelGGL(pgx_scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, s, 1, (const void *)in, n, (uint64_t)0, (const uint64_t *)tmp, nb, out, (uint64_t *)nullptr, raw, (const uint64_t *)nullptr);
    HIPCHECK(hipGetLastError());
}

// scalars the host needs to size the next buffer: device -> a small pinned buffer (a pageable destination makes every such
// copy a staged, blocking transfer) -> caller.  One buffer per host thread.
static void read_scalars(void *dst, const void *dptr, size_t bytes, hipStream_t s) {
    static thread_local void *pin = nullptr;
    if (!pin) HIPCHECK(hipHostMalloc(&pin, 512, hipHostMallocPortable));
    if (bytes > 512) throw Error(PGX_ERR_ARG, "read_scalars: too many bytes");
    HIPCHECK(hipMemcpyAsync(pin, dptr, bytes, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    std::memcpy(dst, pin, bytes);
}
static uint64_t read_u64(const uint64_t *dptr, hipStream_t s) {
    uint64_t v = 0;
    read_scalars(&v, dptr, 8, s);
    return v;
}

static inline unsigned grid_for(uint64_t n, unsigned per_block) {
    uint64_t g = (n + per_block - 1) / per_block;
    if (g == 0) g = 1;
    if (g > 0x7FFFFFFFull) throw Error(PGX_ERR_UNSUPPORTED, "batch too large for one launch");
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------------
// tag pipeline shared by pgx_batch_run and pgx_tag_query_batch
struct TagWork {
    DevBuf run_nums, first_item, seg_off, gbuf, need, scratch_off, scratch, ucount, pos_off, positions, big_list, large_list, scan_tmp, dedup, dd_table,
        single, small_list;
    uint64_t n_positions = 0, n_big = 0;
    // what the last run needed (speculative sizing of the next one, pgx_batch_run): gathered values, positions, list lengths,
    // largest run count on the large list
    uint64_t last_G = 0, last_P = 0, last_small = 0, last_big = 0, last_large = 0, last_largest = 0, last_rep = 0, last_dup = 0;
    bool have_last = false;
    void release() {
        DevBuf *all[] = {&run_nums, &first_item, &seg_off, &gbuf, &need, &scratch_off, &scratch, &ucount, &pos_off, &positions,
                         &big_list, &large_list, &scan_tmp, &dedup, &dd_table, &single, &small_list};
        for (DevBuf *d : all) d->release();
    }
};

static inline uint64_t with_slack(uint64_t v) { return v + v / 4 + 64; }

// Device scalars of the stage, sc[] (zeroed by the caller): [0] big-list length [1] large-list length [2] largest run count on the
// large list [3] total of gathered values G [5] small-list length [6] representatives [7] duplicates [8] positions.
//
// Exact mode (spec == false): the host reads the scalars back where they size the next buffer (three synchronisations).
// Speculative mode: every buffer and grid is sized from the previous run of this batch (+ 25 %), the counts stay on the device
// (kernels take a capacity and a device pointer to the actual count, pgx_tag_kernels.hip), capacity checks raise *d_abort on the
// device, and nothing is read back here: the caller reads all scalars once at the end and repeats the run in exact mode if the
// abort flag came up.  `m` is then the capacity of the per-query arrays and d_m points to the actual number of queries.
template <class Rec>
static void tag_pipeline(const PgxDevImage &img, const pgx_mem *d_mems, const uint64_t *d_qs, const uint64_t *d_qe, uint64_t m, TagWork &w,
                         unsigned long long *d_nover, unsigned long long *sc, hipStream_t s, Rec &&rec, bool spec = false,
                         const uint64_t *d_m = nullptr, uint64_t *d_abort = nullptr) {
    const uint64_t mm = m ? m : 1;
    w.run_nums.ensure(mm * 8);
    w.first_item.ensure(mm * 8);
    w.seg_off.ensure((m + 1) * 8);
    w.need.ensure(mm * 8);
    w.scratch_off.ensure((m + 1) * 8);
    w.ucount.ensure(mm * 8);
    w.pos_off.ensure((m + 1) * 8);
    w.big_list.ensure(mm * 8);
    w.large_list.ensure(mm * 8);
    w.single.ensure(mm * 8);
    w.small_list.ensure(mm * 8);
    const uint64_t *u_sc = reinterpret_cast<const uint64_t *>(sc);
    const uint64_t *dn_small = spec ? u_sc + 5 : nullptr, *dn_big = spec ? u_sc + 0 : nullptr, *dn_large = spec ? u_sc + 1 : nullptr;
    const uint64_t *dn_rep = spec ? u_sc + 6 : nullptr, *dn_dup = spec ? u_sc + 7 : nullptr;
    const uint64_t *ab = spec ? d_abort : nullptr;
    if (!spec) d_m = nullptr; // exact mode: m is the count
    auto fixed_grid = [](uint64_t n, unsigned per_block, unsigned max_blocks) { return (unsigned)std::min<uint64_t>(std::max<uint64_t>((n + per_block - 1) / per_block, 1), max_blocks); };
    if (m) {
        const unsigned g = spec ? fixed_grid(m, PGX_TAG_LOCATE_THREADS, 8192) : grid_for(m, PGX_TAG_LOCATE_THREADS);
        hipLaunchKernelGGL(pgx_tag_locate_kernel, dim3(g), dim3(PGX_TAG_LOCATE_THREADS), 0, s, img, d_mems, d_qs, d_qe, m, d_m, ab,
                           w.run_nums.as<uint64_t>(), w.first_item.as<uint64_t>(), w.need.as<uint64_t>(), w.big_list.as<uint64_t>(), w.large_list.as<uint64_t>(),
                           sc, sc + 1, w.single.as<uint64_t>(), w.ucount.as<uint64_t>(), d_nover, w.small_list.as<uint64_t>(), sc + 5);
        HIPCHECK(hipGetLastError());
    }
    scan_excl(5, w.run_nums.p, m, 0, w.seg_off.as<uint64_t>(), w.scan_tmp, s, reinterpret_cast<uint64_t *>(sc + 3), d_m); // single runs: no segment
    uint64_t G, nbig, nlarge, nsmall, largest;
    if (!spec) {
        uint64_t hv[6] = {0, 0, 0, 0, 0, 0};
        read_scalars(hv, sc, 48, s);
        G = hv[3]; nbig = hv[0]; nlarge = hv[1]; nsmall = hv[5]; largest = hv[2];
    if (std::getenv("PGX_DEBUG_COUNTERS")) std::fprintf(stderr, "[pgx] tag stage: m %llu big %llu large %llu largest %llu G %llu small %llu\n", (unsigned long long)m,
                                                            (unsigned long long)nbig, (unsigned long long)nlarge, (unsigned long long)largest, (unsigned long long)G, (unsigned long long)nsmall);
    } else { // capacities from the previous run; the device checks what it can before anything is written through them
        G = with_slack(w.last_G); nbig = with_slack(w.last_big); nlarge = with_slack(w.last_large); nsmall = with_slack(w.last_small);
        // the large path sorts in dynamic LDS sized for the largest run count: twice the last one (a power of two), at most the
        // workgroup capacity -- a larger query aborts the speculative run (the caller never speculates beyond that capacity)
        uint64_t p2 = 64;
        while (p2 < 2 * w.last_largest && p2 < PGX_SORT_WG_LDS_CAP) p2 <<= 1;
        largest = p2;
        // [bit 0] gathered values, [1] large list, [2] largest run count, [3] big list
        hipLaunchKernelGGL(pgx_spec_check_kernel, dim3(1), dim3(64), 0, s, u_sc + 3, G, u_sc + 1, nlarge, u_sc + 2, largest, u_sc + 0, nbig, d_abort);
        hipLaunchKernelGGL(pgx_spec_check_kernel, dim3(1), dim3(64), 0, s, u_sc + 5, nsmall, (const uint64_t *)nullptr, (uint64_t)0, (const uint64_t *)nullptr,
                           (uint64_t)0, (const uint64_t *)nullptr, (uint64_t)0, d_abort);
        HIPCHECK(hipGetLastError());
        if (largest > PGX_SORT_WG_LDS_CAP) throw Error(PGX_ERR_ARG, "speculative tag stage with a query beyond the LDS sort capacity"); // (the caller never asks for this)
    }
    uint64_t S = 0; // global sort scratch: only queries with more than PGX_SORT_WG_LDS_CAP runs need any (rare: one more scan then)
    if (largest > PGX_SORT_WG_LDS_CAP) {
        scan_excl(1, w.need.p, m, 0, w.scratch_off.as<uint64_t>(), w.scan_tmp, s);
        S = read_u64(w.scratch_off.as<uint64_t>() + m, s);
    }
    w.n_big = nbig;
    rec(0);
    w.gbuf.ensure((G ? G : 1) * 8);
    w.scratch.ensure((S ? S : 1) * 8);
    if (nsmall) { // queries with 2 .. 16 runs (single runs were answered by the locate kernel)
        const unsigned g = spec ? fixed_grid(nsmall, 16, 16384) : grid_for(nsmall, 16);
        hipLaunchKernelGGL(pgx_tag_small_kernel, dim3(g), dim3(256), 0, s, img, (const uint64_t *)w.small_list.as<uint64_t>(), nsmall, dn_small, ab,
                           w.run_nums.as<uint64_t>(), w.first_item.as<uint64_t>(), w.seg_off.as<uint64_t>(), w.gbuf.as<uint64_t>(),
                           w.ucount.as<uint64_t>(), d_nover);
        HIPCHECK(hipGetLastError());
    }
    rec(1);
    if (nbig) {
        const unsigned g = spec ? fixed_grid(nbig, 4, 8192) : grid_for(nbig, 4);
        hipLaunchKernelGGL(pgx_tag_gather_kernel, dim3(g), dim3(256), 0, s, img, (const uint64_t *)w.big_list.as<uint64_t>(), nbig, dn_big, ab,
                           w.run_nums.as<uint64_t>(), w.first_item.as<uint64_t>(), w.seg_off.as<uint64_t>(), w.gbuf.as<uint64_t>(), d_nover);
        hipLaunchKernelGGL(pgx_tag_sort_unique_kernel, dim3(g), dim3(256), 0, s, (const uint64_t *)w.big_list.as<uint64_t>(), nbig, dn_big, ab,
                           w.run_nums.as<uint64_t>(), w.seg_off.as<uint64_t>(), w.gbuf.as<uint64_t>(), w.ucount.as<uint64_t>());
        HIPCHECK(hipGetLastError());
    }
    uint64_t nrep = 0, ndup = 0;
    if (nlarge) {
        uint64_t p2max = 64;
        while (p2max < largest && p2max < PGX_SORT_WG_LDS_CAP) p2max <<= 1;
        const size_t lds = (size_t)p2max * 8; // smaller segments -> more workgroups per CU
        // opt in to > 64 KiB of dynamic LDS (per device; cheap enough to repeat)
        HIPCHECK(hipFuncSetAttribute((const void *)pgx_tag_sort_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(PGX_SORT_WG_LDS_CAP * 8)));
        // identical large queries are grouped on the device (pgx_tag_dedup_kernel): representatives are sorted, duplicates copy
        uint64_t tcap = 64;
        while (tcap < 2 * nlarge) tcap <<= 1;
        w.dd_table.ensure(tcap * 8);
        w.dedup.ensure(3 * nlarge * 8); // [representatives | (duplicate, representative) pairs]
        uint64_t *d_reps = w.dedup.as<uint64_t>(), *d_pairs = d_reps + nlarge;
        HIPCHECK(hipMemsetAsync(w.dd_table.p, 0, tcap * 8, s));
        hipLaunchKernelGGL(pgx_tag_dedup_kernel, dim3(fixed_grid(nlarge, 256, 1024)), dim3(256), 0, s, (const uint64_t *)w.large_list.as<uint64_t>(), nlarge, dn_large, ab,
                           (const uint64_t *)w.first_item.as<uint64_t>(), (const uint64_t *)w.run_nums.as<uint64_t>(), w.dd_table.as<unsigned long long>(), tcap - 1,
                           d_reps, sc + 6, d_pairs, sc + 7);
        HIPCHECK(hipGetLastError());
        if (!spec) {
            uint64_t rd[2] = {0, 0};
            read_scalars(rd, sc + 6, 16, s);
            nrep = rd[0]; ndup = rd[1];
    if (std::getenv("PGX_DEBUG_COUNTERS")) std::fprintf(stderr, "[pgx] tag stage: representatives %llu duplicates %llu\n", (unsigned long long)nrep, (unsigned long long)ndup);
        } else { nrep = nlarge; ndup = nlarge; } // (capacities: the lists cannot be longer than the large list)
        if (nrep) {
            hipLaunchKernelGGL(pgx_tag_gather_kernel, dim3(spec ? fixed_grid(nrep, 4, 8192) : grid_for(nrep, 4)), dim3(256), 0, s, img, (const uint64_t *)d_reps, nrep,
                               dn_rep, ab, w.run_nums.as<uint64_t>(), w.first_item.as<uint64_t>(), w.seg_off.as<uint64_t>(), w.gbuf.as<uint64_t>(), d_nover);
            hipLaunchKernelGGL(pgx_tag_sort_large_kernel, dim3(spec ? fixed_grid(nrep, 1, 2048) : grid_for(nrep, 1)), dim3(1024), lds, s, (const uint64_t *)d_reps, nrep,
                               dn_rep, ab, w.run_nums.as<uint64_t>(), w.seg_off.as<uint64_t>(), w.gbuf.as<uint64_t>(), w.scratch.as<uint64_t>(),
                               w.scratch_off.as<uint64_t>(), w.ucount.as<uint64_t>());
        }
        if (ndup)
            hipLaunchKernelGGL(pgx_tag_copy_dups_kernel, dim3(spec ? fixed_grid(ndup, 256, 4096) : grid_for(ndup, 256)), dim3(256), 0, s, (const uint64_t *)d_pairs, ndup,
                               dn_dup, ab, img.n_tag_items, w.first_item.as<uint64_t>(), w.run_nums.as<uint64_t>(), w.seg_off.as<uint64_t>(),
                               w.gbuf.as<uint64_t>(), w.ucount.as<uint64_t>(), d_nover);
        HIPCHECK(hipGetLastError());
    }
    scan_excl(1, w.ucount.p, m, 0, w.pos_off.as<uint64_t>(), w.scan_tmp, s, reinterpret_cast<uint64_t *>(sc + 8), d_m);
    uint64_t P;
    if (!spec) {
        w.n_positions = read_u64(reinterpret_cast<const uint64_t *>(sc + 8), s);
        P = w.n_positions;
    } else {
        P = with_slack(w.last_P);
        hipLaunchKernelGGL(pgx_spec_check_kernel, dim3(1), dim3(64), 0, s, u_sc + 8, P, (const uint64_t *)nullptr, (uint64_t)0, (const uint64_t *)nullptr, (uint64_t)0,
                           (const uint64_t *)nullptr, (uint64_t)0, d_abort);
    }
    w.positions.ensure((P ? P : 1) * 8);
    if (m) {
        // every query is on exactly one list: single (thread per query), small, big, large (16 lanes per query up to
        // PGX_TAG_COMPACT_SMALL unique values, one workgroup per query beyond)
        const uint64_t *lists[3] = {w.small_list.as<uint64_t>(), w.big_list.as<uint64_t>(), w.large_list.as<uint64_t>()};
        const uint64_t counts[3] = {nsmall, nbig, nlarge};
        const uint64_t *dcounts[3] = {dn_small, dn_big, dn_large};
        for (int li = 0; li < 3; li++)
            if (counts[li])
                hipLaunchKernelGGL(pgx_tag_compact_kernel, dim3(spec ? fixed_grid(counts[li], 16, 16384) : grid_for(counts[li], 16)), dim3(256), 0, s, lists[li], counts[li],
                                   dcounts[li], ab, w.ucount.as<uint64_t>(), w.seg_off.as<uint64_t>(), w.gbuf.as<uint64_t>(), w.pos_off.as<uint64_t>(),
                                   w.positions.as<uint64_t>(), (uint64_t)PGX_TAG_COMPACT_SMALL);
        hipLaunchKernelGGL(pgx_tag_compact_single_kernel, dim3(spec ? fixed_grid(m, 256, 16384) : grid_for(m, 256)), dim3(256), 0, s, m, d_m, ab,
                           (const uint64_t *)w.run_nums.as<uint64_t>(), (const uint64_t *)w.single.as<uint64_t>(), (const uint64_t *)w.pos_off.as<uint64_t>(),
                           w.positions.as<uint64_t>());
        if (nbig)
            hipLaunchKernelGGL(pgx_tag_compact_list_kernel, dim3(spec ? fixed_grid(nbig, 1, 4096) : grid_for(nbig, 1)), dim3(256), 0, s, (const uint64_t *)w.big_list.as<uint64_t>(), nbig,
                               dn_big, ab, w.ucount.as<uint64_t>(), w.seg_off.as<uint64_t>(), w.gbuf.as<uint64_t>(), w.pos_off.as<uint64_t>(),
                               w.positions.as<uint64_t>(), (uint64_t)PGX_TAG_COMPACT_SMALL);
        if (nlarge)
            hipLaunchKernelGGL(pgx_tag_compact_list_kernel, dim3(spec ? fixed_grid(nlarge, 1, 4096) : grid_for(nlarge, 1)), dim3(256), 0, s,
                               (const uint64_t *)w.large_list.as<uint64_t>(), nlarge, dn_large, ab, w.ucount.as<uint64_t>(), w.seg_off.as<uint64_t>(),
                               w.gbuf.as<uint64_t>(), w.pos_off.as<uint64_t>(), w.positions.as<uint64_t>(), (uint64_t)PGX_TAG_COMPACT_SMALL);
        HIPCHECK(hipGetLastError());
    }
    rec(2);
    if (!spec) { // what the next run of this batch may assume
        w.last_G = G; w.last_P = P; w.last_small = nsmall; w.last_big = nbig; w.last_large = nlarge; w.last_largest = largest;
        w.last_rep = nrep; w.last_dup = ndup;
        w.have_last = true;
    }
}

// ------------------------------------------------------------------------------------------
// locate path (pgx_locate_kernels.hip)
static pgx_device_image *locate_image(pgx_index *h, int device) {
    if (!h->has_rank) throw Error(PGX_ERR_ARG, "locate: index opened without an r-index");
    pgx_device_image *d = device_image(h, device);
    std::lock_guard<std::mutex> lock(g_image_mutex);
    if (d->has_loc) return d;
    build_locate_image(h->ri, h->loc);
    const LocHostImage &m = h->loc;
    upload(d->rstart, m.rstart.data(), m.rstart.size() * 8);
    upload(d->rsamp, m.rsamp.data(), m.rsamp.size() * 8);
    upload(d->rdir, m.rdir.data(), m.rdir.size() * 4);
    upload(d->lpos, m.lpos.data(), m.lpos.size() * 8);
    upload(d->lnext, m.lnext.data(), m.lnext.size() * 8);
    upload(d->ldir, m.ldir.data(), m.ldir.size() * 4);
    PgxLocImage &g = d->loc;
    g.rstart = d->rstart.as<uint64_t>(); g.rsamp = d->rsamp.as<uint64_t>(); g.rdir = d->rdir.as<uint32_t>();
    g.lpos = d->lpos.as<uint64_t>(); g.lnext = d->lnext.as<uint64_t>(); g.ldir = d->ldir.as<uint32_t>();
    g.n = m.consts.n; g.n_runs = m.consts.n_runs; g.n_last = m.consts.n_last; g.max_length = m.consts.max_length;
    g.rdir_entries = m.consts.rdir_entries; g.ldir_entries = m.consts.ldir_entries;
    g.rdir_shift = m.consts.rdir_shift; g.ldir_shift = m.consts.ldir_shift;
    d->has_loc = true;
    return d;
}

extern "C" pgx_status pgx_locate_next_batch(pgx_index *h, int device, const uint64_t *prev, uint64_t n, uint64_t *out) {
    PGX_GUARD_BEGIN
    if (!h || (n && (!prev || !out))) throw Error(PGX_ERR_ARG, "pgx_locate_next_batch: null argument");
    pgx_device_image *d = locate_image(h, device);
    if (!n) return PGX_OK;
    DevBuf di, dout;
    try {
        di.ensure(n * 8); dout.ensure(n * 8);
        HIPCHECK(hipMemcpy(di.p, prev, n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(pgx_locate_next_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nullptr, d->loc, di.as<uint64_t>(), n,
                           dout.as<uint64_t>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
    } catch (...) {
        di.release(); dout.release();
        throw;
    }
    di.release(); dout.release();
    return PGX_OK;
    PGX_GUARD_END
}

// the walk + optional segmented sort-unique; results stay in w_vals (values) / h_off (host offsets)
static void locate_core(pgx_index *h, pgx_device_image *d, const uint64_t *first, const uint64_t *last, uint64_t n, uint32_t flags,
                        std::vector<uint64_t> &h_off, DevBuf &vals_out, uint64_t &n_vals_out) {
    const uint64_t bwt_n = d->loc.n;
    std::vector<uint64_t> cnt(n), voff(n + 1);
    voff[0] = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (last[i] < first[i]) cnt[i] = 0;
        else {
            if (last[i] >= bwt_n) throw Error(PGX_ERR_ARG, "pgx_locate_batch: range " + std::to_string(i) + " ends beyond the BWT");
            cnt[i] = last[i] - first[i] + 1;
        }
        voff[i + 1] = voff[i] + cnt[i];
    }
    const uint64_t V = voff[n];
    DevBuf dqs, dqe, drun0, dnp, dpoff, dvoff, dcnt, scan_tmp, dlist, dneed, dsoff, dscratch, ducount, duoff;
    DevBuf *all[] = {&dqs, &dqe, &drun0, &dnp, &dpoff, &dvoff, &dcnt, &scan_tmp, &dlist, &dneed, &dsoff, &dscratch, &ducount, &duoff};
    DevBuf gbuf;
    try {
        hipStream_t s = nullptr;
        dqs.ensure(n * 8); dqe.ensure(n * 8); drun0.ensure(n * 8); dnp.ensure(n * 8); dpoff.ensure((n + 1) * 8); dvoff.ensure((n + 1) * 8);
        HIPCHECK(hipMemcpy(dqs.p, first, n * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(dqe.p, last, n * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(dvoff.p, voff.data(), (n + 1) * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(pgx_locate_plan_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d->loc, dqs.as<uint64_t>(), dqe.as<uint64_t>(), n,
                           drun0.as<uint64_t>(), dnp.as<uint64_t>());
        HIPCHECK(hipGetLastError());
        scan_excl(1, dnp.p, n, 0, dpoff.as<uint64_t>(), scan_tmp, s);
        const uint64_t n_pieces = read_u64(dpoff.as<uint64_t>() + n, s);
        gbuf.ensure((V ? V : 1) * 8);
        if (n_pieces) {
            hipLaunchKernelGGL(pgx_locate_walk_kernel, dim3(grid_for(n_pieces, 256)), dim3(256), 0, s, d->loc, dqs.as<uint64_t>(),
                               dqe.as<uint64_t>(), n, drun0.as<uint64_t>(), dpoff.as<uint64_t>(), n_pieces, dvoff.as<uint64_t>(),
                               (flags & PGX_LOCATE_SEQ_IDS) ? 1 : 0, gbuf.as<uint64_t>());
            HIPCHECK(hipGetLastError());
        }
        if (!(flags & PGX_LOCATE_UNIQUE)) {
            HIPCHECK(hipStreamSynchronize(s));
            h_off = voff;
            vals_out = gbuf; gbuf = DevBuf();
            n_vals_out = V;
        } else {
            // segmented sort + unique with the tag path's kernels: one wave per segment up to 2048 values, one
            // 1024-thread workgroup beyond (LDS up to 16384 values, global scratch above)
            std::vector<uint64_t> wave_list, wg_list, need(n, 0), soff(n + 1, 0);
            uint64_t max_cnt = 0;
            for (uint64_t i = 0; i < n; i++) {
                if (cnt[i] == 0) continue;
                if (cnt[i] <= PGX_SORT_LDS_CAP) wave_list.push_back(i);
                else {
                    wg_list.push_back(i);
                    max_cnt = std::max(max_cnt, cnt[i]);
                    uint64_t p2 = 64;
                    while (p2 < cnt[i]) p2 <<= 1;
                    if (p2 > PGX_SORT_WG_LDS_CAP) need[i] = p2;
                }
            }
            for (uint64_t i = 0; i < n; i++) soff[i + 1] = soff[i] + need[i];
            dcnt.ensure(n * 8); ducount.ensure(n * 8); duoff.ensure((n + 1) * 8);
            dlist.ensure((n ? n : 1) * 8); dsoff.ensure((n + 1) * 8); dscratch.ensure((soff[n] ? soff[n] : 1) * 8);
            HIPCHECK(hipMemcpy(dcnt.p, cnt.data(), n * 8, hipMemcpyHostToDevice));
            HIPCHECK(hipMemset(ducount.p, 0, n * 8)); // empty ranges are on no list
            HIPCHECK(hipMemcpy(dsoff.p, soff.data(), (n + 1) * 8, hipMemcpyHostToDevice));
            uint64_t *d_wave = dlist.as<uint64_t>(), *d_wg = d_wave + wave_list.size();
            if (!wave_list.empty()) HIPCHECK(hipMemcpy(d_wave, wave_list.data(), wave_list.size() * 8, hipMemcpyHostToDevice));
            if (!wg_list.empty()) HIPCHECK(hipMemcpy(d_wg, wg_list.data(), wg_list.size() * 8, hipMemcpyHostToDevice));
            if (!wave_list.empty())
                hipLaunchKernelGGL(pgx_tag_sort_unique_kernel, dim3(grid_for(wave_list.size(), 4)), dim3(256), 0, s, (const uint64_t *)d_wave,
                                   (uint64_t)wave_list.size(), (const uint64_t *)nullptr, (const uint64_t *)nullptr, dcnt.as<uint64_t>(), dvoff.as<uint64_t>(),
                                   gbuf.as<uint64_t>(), ducount.as<uint64_t>());
            if (!wg_list.empty()) {
                uint64_t p2max = 64;
                while (p2max < max_cnt && p2max < PGX_SORT_WG_LDS_CAP) p2max <<= 1;
                HIPCHECK(hipFuncSetAttribute((const void *)pgx_tag_sort_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(PGX_SORT_WG_LDS_CAP * 8)));
                hipLaunchKernelGGL(pgx_tag_sort_large_kernel, dim3(grid_for(wg_list.size(), 1)), dim3(1024), (size_t)p2max * 8, s,
                                   (const uint64_t *)d_wg, (uint64_t)wg_list.size(), (const uint64_t *)nullptr, (const uint64_t *)nullptr, dcnt.as<uint64_t>(),
                                   dvoff.as<uint64_t>(), gbuf.as<uint64_t>(), dscratch.as<uint64_t>(), dsoff.as<uint64_t>(), ducount.as<uint64_t>());
            }
            HIPCHECK(hipGetLastError());
            scan_excl(1, ducount.p, n, 0, duoff.as<uint64_t>(), scan_tmp, s);
            const uint64_t U = read_u64(duoff.as<uint64_t>() + n, s);
            vals_out.ensure((U ? U : 1) * 8);
            hipLaunchKernelGGL(pgx_tag_compact_kernel, dim3(grid_for(n, 16)), dim3(256), 0, s, (const uint64_t *)nullptr, n, (const uint64_t *)nullptr, (const uint64_t *)nullptr,
                               ducount.as<uint64_t>(), dvoff.as<uint64_t>(), gbuf.as<uint64_t>(), duoff.as<uint64_t>(), vals_out.as<uint64_t>(), ~0ull);
            HIPCHECK(hipGetLastError());
            h_off.resize(n + 1);
            HIPCHECK(hipMemcpy(h_off.data(), duoff.p, (n + 1) * 8, hipMemcpyDeviceToHost));
            n_vals_out = U;
        }
    } catch (...) {
        for (DevBuf *b : all) b->release();
        gbuf.release();
        throw;
    }
    for (DevBuf *b : all) b->release();
    gbuf.release();
}

static void locate_check_supported(const pgx_index *h, const char *who) {
    if (h->mode == PGX_MODE_COMPAT && h->ri.encoded && !h->ri.hasN)
        throw Error(PGX_ERR_UNSUPPORTED, std::string(who) + ": the reference's encoded run scan skips six header varints where five were "
                                         "written on an index without N (src/r-index.cpp:83-88); open the index in PGX_MODE_STRICT");
}

extern "C" pgx_status pgx_locate_batch(pgx_index *h, int device, const uint64_t *first, const uint64_t *last, uint64_t n, uint32_t flags,
                                       uint64_t *val_offsets, uint64_t *values, uint64_t values_cap) {
    PGX_GUARD_BEGIN
    if (!h || !val_offsets || (n && (!first || !last))) throw Error(PGX_ERR_ARG, "pgx_locate_batch: null argument");
    if (flags & ~(PGX_LOCATE_SEQ_IDS | PGX_LOCATE_UNIQUE)) throw Error(PGX_ERR_ARG, "pgx_locate_batch: unknown flag");
    if (!h->has_rank) throw Error(PGX_ERR_ARG, "pgx_locate_batch: index opened without an r-index");
    locate_check_supported(h, "pgx_locate_batch");
    pgx_device_image *d = locate_image(h, device);
    val_offsets[0] = 0;
    if (!n) return PGX_OK;
    std::vector<uint64_t> off;
    DevBuf vals;
    uint64_t nv = 0;
    try {
        locate_core(h, d, first, last, n, flags, off, vals, nv);
        std::copy(off.begin(), off.end(), val_offsets);
        if (values) {
            if (values_cap < nv) throw Error(PGX_ERR_ARG, "pgx_locate_batch: values_cap too small");
            if (nv) HIPCHECK(hipMemcpy(values, vals.p, nv * 8, hipMemcpyDeviceToHost));
        }
    } catch (...) {
        vals.release();
        throw;
    }
    vals.release();
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_decompress_sa(pgx_index *h, int device, uint32_t flags, uint64_t *out) {
    PGX_GUARD_BEGIN
    if (!h || !out) throw Error(PGX_ERR_ARG, "pgx_decompress_sa: null argument");
    if (flags & ~PGX_LOCATE_SEQ_IDS) throw Error(PGX_ERR_ARG, "pgx_decompress_sa: unknown flag");
    pgx_device_image *d = locate_image(h, device); // run boundaries come from the parsed blocks, not from the reference's scan
    if (!d->loc.n) return PGX_OK;
    const uint64_t first = 0, last = d->loc.n - 1;
    std::vector<uint64_t> off;
    DevBuf vals;
    uint64_t nv = 0;
    try {
        locate_core(h, d, &first, &last, 1, flags, off, vals, nv);
        HIPCHECK(hipMemcpy(out, vals.p, nv * 8, hipMemcpyDeviceToHost));
    } catch (...) {
        vals.release();
        throw;
    }
    vals.release();
    return PGX_OK;
    PGX_GUARD_END
}

// ------------------------------------------------------------------------------------------
// merge_tags (pgx_merge_kernels.hip)
static void merge_tags_core(const char *ri_path, const char *const *tag_paths, uint32_t n_files, const uint32_t *seq_to_file, uint64_t n_seq,
                            int device, const char *out_path, uint64_t max_node_floor, uint32_t opts) {
    if (!ri_path || !tag_paths || !seq_to_file || !out_path) throw Error(PGX_ERR_ARG, "pgx_merge_tags: null argument");
    if (opts & ~PGX_MERGE_REFERENCE_RUNS) throw Error(PGX_ERR_ARG, "pgx_merge_tags: unknown flag");
    if (n_files == 0 || n_files > 250) throw Error(PGX_ERR_ARG, "pgx_merge_tags: between 1 and 250 tag files");
    // only the locate side of the index is needed: parse the file, no rank image
    std::unique_ptr<pgx_index, void (*)(pgx_index *)> guard(new pgx_index(), pgx_index_close);
    pgx_index *h = guard.get();
    {
        std::vector<uint8_t> f;
        try { f = read_whole_file(ri_path); }
        catch (const Error &) { throw Error(PGX_ERR_IO, std::string("Cannot open r-index: ") + ri_path); }
        h->ri.parse(f.data(), f.size());
        std::memset(&h->img.consts, 0, sizeof h->img.consts);
        h->mode = PGX_MODE_STRICT;
        h->has_rank = true;
    }
    // the tag streams are parsed by host threads (one per file) while the device computes the document array
    struct Stream { std::vector<uint64_t> st, vl; std::string err; };
    std::vector<Stream> streams(n_files);
    std::vector<std::thread> parsers;
    for (uint32_t f = 0; f < n_files; f++)
        if (!tag_paths[f]) throw Error(PGX_ERR_ARG, "pgx_merge_tags: null tag path"); // before any thread exists
    // joins whatever was started, also when starting a later thread throws (a joinable std::thread must not be destroyed)
    struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner{parsers};
    parsers.reserve(n_files);
    for (uint32_t f = 0; f < n_files; f++) {
        parsers.emplace_back([&streams, tag_paths, f]() {
            Stream &o = streams[f];
            try {
                std::vector<uint8_t> raw = read_whole_file(tag_paths[f]);
                uint64_t loc = 0;
                if (raw.size() >= 8) { // int_vector<8> header (bit count) of sdsl::int_vector_buffer<8>, merge_tags.cpp:207
                    uint64_t bits = 0;
                    std::memcpy(&bits, raw.data(), 8);
                    if (bits == (raw.size() - 8) * 8) loc = 8;
                }
                o.st.assign(1, 0);
                while (loc < raw.size()) {
                    const uint64_t v = bytecode_read(raw.data(), raw.size(), loc, "tag run");
                    const uint64_t len = (v >> 11) & 0x1FF; // decode_run, length_bits = 9 (src/tag_arrays.cpp:59-70)
                    if (!len) continue;
                    o.vl.push_back((v & 0x7FF) | ((v >> 20) << 11)); // offset | rev << 10 | node << 11
                    o.st.push_back(o.st.back() + len);
                }
            } catch (const std::exception &e) { o.err = e.what(); }
        });
    }
    const uint64_t n = h->ri.sequence_size, tot = h->ri.C.size() > 1 ? h->ri.C[1] - h->ri.C[0] : 0;
    if (n_seq != tot) throw Error(PGX_ERR_ARG, "pgx_merge_tags: seq_to_file has " + std::to_string(n_seq) + " entries, the index holds " +
                                                   std::to_string(tot) + " sequences");
    pgx_device_image *d = locate_image(h, device);
    DevBuf da, file_of, tags, rank, scan_tmp, s2f, ctr, rstart, rval, expanded, flags, out_val, out_start;
    DevBuf *all[] = {&da, &file_of, &tags, &rank, &scan_tmp, &s2f, &ctr, &rstart, &rval, &expanded, &flags, &out_val, &out_start};
    std::vector<uint64_t> h_val, h_start;
    try {
        hipStream_t s = nullptr;
        // 1. document array of the whole BWT
        {
            const uint64_t first = 0, last = n ? n - 1 : 0;
            std::vector<uint64_t> off;
            uint64_t nv = 0;
            if (n) locate_core(h, d, &first, &last, 1, PGX_LOCATE_SEQ_IDS, off, da, nv);
        }
        // 2. file of every position
        file_of.ensure(n ? n : 1); tags.ensure((n ? n : 1) * 8); rank.ensure((n + 1) * 8); s2f.ensure((n_seq ? n_seq : 1) * 4); ctr.ensure(64);
        HIPCHECK(hipMemsetAsync(ctr.p, 0, 64, s));
        HIPCHECK(hipMemsetAsync(tags.p, 0, (n ? n : 1) * 8, s));
        if (n_seq) HIPCHECK(hipMemcpyAsync(s2f.p, seq_to_file, n_seq * 4, hipMemcpyHostToDevice, s));
        if (n) {
            hipLaunchKernelGGL(pgx_mt_file_of_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, da.as<uint64_t>(), n, n_seq, s2f.as<uint32_t>(), n_files,
                               file_of.as<uint8_t>(), ctr.as<unsigned long long>());
            HIPCHECK(hipGetLastError());
        }
        if (read_u64(ctr.as<uint64_t>(), s)) throw Error(PGX_ERR_ARG, "pgx_merge_tags: seq_to_file names a file index >= n_files");
        da.release();
        // 3. per file: expanded stream, rank of its positions, gather
        for (uint32_t f = 0; f < n_files; f++) {
            parsers[f].join();
            if (!streams[f].err.empty()) throw Error(PGX_ERR_FORMAT, std::string(tag_paths[f]) + ": " + streams[f].err);
            const std::vector<uint64_t> &st = streams[f].st, &vl = streams[f].vl;
            const uint64_t nr = vl.size(), total = st.back();
            scan_excl(3, file_of.p, n, f, rank.as<uint64_t>(), scan_tmp, s);
            const uint64_t have = read_u64(rank.as<uint64_t>() + n, s);
            if (have != total)
                throw Error(PGX_ERR_FORMAT, std::string("pgx_merge_tags: ") + tag_paths[f] + " holds " + std::to_string(total) + " tags, the BWT has " +
                                                std::to_string(have) + " positions of its sequences");
            if (!total) continue;
            rstart.ensure((nr + 1) * 8); rval.ensure(nr * 8); expanded.ensure(total * 8);
            HIPCHECK(hipMemcpyAsync(rstart.p, st.data(), (nr + 1) * 8, hipMemcpyHostToDevice, s));
            HIPCHECK(hipMemcpyAsync(rval.p, vl.data(), nr * 8, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(pgx_mt_expand_kernel, dim3(grid_for(nr, 256)), dim3(256), 0, s, rstart.as<uint64_t>(), rval.as<uint64_t>(), nr,
                               expanded.as<uint64_t>());
            hipLaunchKernelGGL(pgx_mt_gather_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, file_of.as<uint8_t>(), f, rank.as<uint64_t>(),
                               expanded.as<uint64_t>(), total, n, tags.as<uint64_t>());
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipStreamSynchronize(s)); // st / vl are host vectors read by the async copies
        }
        // 4. run-length encode
        uint64_t n_out = 0;
        if (n) {
            flags.ensure(n);
            hipLaunchKernelGGL(pgx_mt_flags_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, tags.as<uint64_t>(), n, n_seq, flags.as<uint8_t>());
            scan_excl(4, flags.p, n, 0, rank.as<uint64_t>(), scan_tmp, s);
            n_out = read_u64(rank.as<uint64_t>() + n, s);
            out_val.ensure(n_out * 8); out_start.ensure(n_out * 8);
            hipLaunchKernelGGL(pgx_mt_compact_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, tags.as<uint64_t>(), flags.as<uint8_t>(), rank.as<uint64_t>(),
                               n, n_seq, out_val.as<uint64_t>(), out_start.as<uint64_t>());
            HIPCHECK(hipGetLastError());
            h_val.resize(n_out); h_start.resize(n_out + 1);
            HIPCHECK(hipMemcpy(h_val.data(), out_val.p, n_out * 8, hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(h_start.data(), out_start.p, n_out * 8, hipMemcpyDeviceToHost));
            h_start[n_out] = n;
        }
        for (uint64_t i = 0; i < n_out; i++) h_start[i] = h_start[i + 1] - h_start[i]; // lengths
        if (opts & PGX_MERGE_REFERENCE_RUNS) {
            // the reference counts a merged run in a uint16_t (std::pair<pos_t, uint16_t>, src/merge_tags.cpp:282,346,394-398,625) and
            // adds the pieces of a run that crosses a 500-run job in the same type (:776-777): what reaches
            // append_compact_run_streamed is the maximal run's length mod 65 536, and a length of 0 writes nothing (tag_arrays.cpp:959)
            uint64_t w = 0;
            for (uint64_t i = 0; i < n_out; i++) {
                const uint64_t l16 = h_start[i] & 0xFFFFull;
                if (!l16) continue;
                h_val[w] = h_val[i]; h_start[w++] = l16;
            }
            h_val.resize(w); h_start.resize(w + 1);
        }
    } catch (...) {
        for (DevBuf *b : all) b->release();
        throw;
    }
    for (DevBuf *b : all) b->release();
    write_compact_tags(out_path, h_val.data(), h_start.data(), h_val.size(), max_node_floor);
}

extern "C" pgx_status pgx_merge_tags(const char *ri_path, const char *const *tag_paths, uint32_t n_files, const uint32_t *seq_to_file,
                                     uint64_t n_seq, int device, const char *out_path) {
    PGX_GUARD_BEGIN
    merge_tags_core(ri_path, tag_paths, n_files, seq_to_file, n_seq, device, out_path, 0, 0);
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_merge_tags_ex(const char *ri_path, const char *const *tag_paths, uint32_t n_files, const uint32_t *seq_to_file,
                                        uint64_t n_seq, int device, const char *out_path, uint32_t flags) {
    PGX_GUARD_BEGIN
    merge_tags_core(ri_path, tag_paths, n_files, seq_to_file, n_seq, device, out_path, 0, flags);
    return PGX_OK;
    PGX_GUARD_END
}

// first tag of a per-chromosome stream (FileReader::get_first_tag, src/merge_tags.cpp:205-232): its node id
static uint64_t first_tag_node(const char *path) {
    std::vector<uint8_t> raw = read_whole_file(path);
    uint64_t loc = 0;
    if (raw.size() >= 8) { // int_vector<8> header (bit count) of sdsl::int_vector_buffer<8>
        uint64_t bits = 0;
        std::memcpy(&bits, raw.data(), 8);
        if (bits == (raw.size() - 8) * 8) loc = 8;
    }
    if (loc >= raw.size()) throw Error(PGX_ERR_FORMAT, std::string(path) + ": empty tag file");
    const uint64_t v = bytecode_read(raw.data(), raw.size(), loc, "tag run");
    return v >> 20; // offset:10 | rev:1 | len:9 | node << 20 (encode_run_length, src/tag_arrays.cpp:28-36)
}

static void merge_tags_gbz_core(const char *gbz_path, const char *ri_path, const char *const *tag_paths, uint32_t n_files, int device,
                                const char *out_path, uint32_t flags) {
    if (!gbz_path || !ri_path || !tag_paths || !out_path || !n_files) throw Error(PGX_ERR_ARG, "pgx_merge_tags_gbz: null argument");
    GbzPaths g;
    try { parse_gbz_paths(gbz_path, g); }
    catch (const Error &e) { if (e.code == PGX_ERR_IO) throw Error(PGX_ERR_IO, std::string("Cannot open graph: ") + gbz_path); throw; }
    // component -> file: the component of the first tag's node of every file (merge_tags.cpp:481-490)
    std::vector<uint32_t> comp_to_file(g.n_components, ~0u);
    for (uint32_t f = 0; f < n_files; f++) {
        if (!tag_paths[f]) throw Error(PGX_ERR_ARG, "pgx_merge_tags_gbz: null tag path");
        const uint64_t node = first_tag_node(tag_paths[f]);
        // a node the graph does not have (0: a stream that opens with a gap run) lands in component 0 like the reference's
        // node_to_comp_map[...] (std::unordered_map::operator[] default-inserts 0, merge_tags.cpp:489)
        const uint32_t c = (node < g.component_of_node.size() && g.component_of_node[node] != ~0u) ? g.component_of_node[node] : 0u;
        if (c >= g.n_components) throw Error(PGX_ERR_FORMAT, "pgx_merge_tags_gbz: the graph has no component");
        if (comp_to_file[c] != ~0u) throw Error(PGX_ERR_FORMAT, std::string(tag_paths[f]) + ": a second tag file for the same graph component");
        comp_to_file[c] = f;
    }
    std::vector<uint32_t> s2f(g.first_node.size());
    for (uint64_t sq = 0; sq < s2f.size(); sq++) {
        const uint64_t node = g.first_node[sq];
        const uint32_t c = node ? g.component_of_node[node] : ~0u;
        if (c == ~0u || comp_to_file[c] == ~0u)
            throw Error(PGX_ERR_FORMAT, "path " + std::to_string(sq) + " of the graph starts in a component without a tag file");
        s2f[sq] = comp_to_file[c];
    }
    merge_tags_core(ri_path, tag_paths, n_files, s2f.data(), s2f.size(), device, out_path, g.max_node_id, flags);
}

extern "C" pgx_status pgx_merge_tags_gbz(const char *gbz_path, const char *ri_path, const char *const *tag_paths, uint32_t n_files, int device,
                                         const char *out_path) {
    PGX_GUARD_BEGIN
    merge_tags_gbz_core(gbz_path, ri_path, tag_paths, n_files, device, out_path, 0);
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_merge_tags_gbz_ex(const char *gbz_path, const char *ri_path, const char *const *tag_paths, uint32_t n_files, int device,
                                            const char *out_path, uint32_t flags) {
    PGX_GUARD_BEGIN
    merge_tags_gbz_core(gbz_path, ri_path, tag_paths, n_files, device, out_path, flags);
    return PGX_OK;
    PGX_GUARD_END
}

// ------------------------------------------------------------------------------------------
struct pgx_chunk { uint64_t r0, r1, slot_base, slots; }; // consecutive reads sharing one pass over the slot buffer

struct pgx_batch {
    pgx_index *h = nullptr;
    pgx_device_image *dimg = nullptr;
    int device = 0;
    hipStream_t own = nullptr; // non-blocking stream of this batch: its copies, and its kernels when the caller names no stream
    // second stream: the dense2 kernel over the reads with a byte outside A C G T, while the two-step kernel runs (pgx_classify_reads_kernel)
    hipStream_t side = nullptr;
    hipEvent_t ev_side[2] = {nullptr, nullptr};
    bool class_valid = false, class_ok = false; // read_flags / side_list / side_count describe the uploaded reads (ok: few enough such reads to list)
    uint64_t side_reads_est = 0;                 // about how many reads the second-stream launch serves (sizes its grid)
    uint64_t n_reads = 0, read_bytes = 0;
    HostBuf h_off[2];                // rebased host copy of the offsets (chunk planning; pinned: its upload runs at link speed), and the one being filled
    int h_off_cur = 0;
    const uint64_t *h_offsets() const { return h_off[h_off_cur].as<uint64_t>(); }
    std::vector<pgx_chunk> chunks;   // plan of the last run (reused while min_len / budget are unchanged)
    bool plan_valid = false, slot_off_valid = false;
    uint64_t slot_off_min_len = 0;
    uint64_t plan_min_len = 0, plan_budget = 0;
    DevBuf reads, offsets;
    // run state
    DevBuf slot_off, slots, mem_count, mem_off, mems, scan_tmp, counters, heavy_list, heavy_scratch, read_flags, side_list, side_count, packed, ovf_base;
    DevBuf up_side_ids, up_side_off, up_side_bytes; // pgx_batch_upload_packed: the listed reads as they arrive
    std::vector<uint64_t> h_side_off;
    hipEvent_t ev_up[2] = {nullptr, nullptr};        // around the device passes of an upload
    float ms_upload_passes = 0;                      // device time of the passes this upload needed before its first find_mems launch (pgx_timing.ms_per_upload adds the run's own)
    uint64_t last_ovf_used = 0; // arena slots the last run handed out (sizes the next arena)
    uint64_t max_read_len = 0; // longest read of the upload (sizes the LDS columns of the packed pairs kernel)
    TagWork tw;
    uint64_t n_mems = 0, n_positions = 0, n_ext = 0, n_tag_overflow = 0;
    bool ran = false, ran_tags = false;
    // speculative sizing (pgx_batch_run): what the last run with these parameters produced
    bool shape_valid = false;
    uint64_t shape_reads = 0, shape_min_len = 0, shape_min_occ = 0, last_mems = 0;
    bool shape_tags = false;
    uint32_t spec_runs = 0, spec_fallbacks = 0;
    // host copies
    HostBuf h_mem_off, h_mems, h_run_nums, h_pos_off, h_positions;
    // timing
    hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // [8]: behind the first find_mems launch; [9]: before it, behind the passes a fresh upload needs
    bool timed = false;
    pgx_timing timing{};
};

static void batch_release(pgx_batch *b) {
    if (!b) return;
    if (hipSetDevice(b->device) == hipSuccess) {
        DevBuf *all[] = {&b->reads, &b->offsets, &b->slot_off, &b->slots, &b->mem_count, &b->mem_off, &b->mems, &b->scan_tmp,
                         &b->counters, &b->heavy_list, &b->heavy_scratch, &b->read_flags, &b->side_list, &b->side_count, &b->packed, &b->ovf_base,
                         &b->up_side_ids, &b->up_side_off, &b->up_side_bytes};
        for (DevBuf *d : all) d->release();
        b->tw.release();
        HostBuf *hb[] = {&b->h_mem_off, &b->h_mems, &b->h_run_nums, &b->h_pos_off, &b->h_positions, &b->h_off[0], &b->h_off[1]};
        for (HostBuf *x : hb) x->release();
        for (auto &e : b->ev)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (b->own) { (void)hipStreamDestroy(b->own); b->own = nullptr; }
        if (b->side) { (void)hipStreamDestroy(b->side); b->side = nullptr; }
        for (auto &e : b->ev_side)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (auto &e : b->ev_up)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
    }
    delete b;
}

extern "C" void pgx_batch_free(pgx_batch *b) { batch_release(b); }

// (re)fill a batch: device buffers only ever grow, so a long-lived batch costs no allocation per call
// offsets: validated, rebased to 0 (host copy for the chunk planner, device copy for the kernels), longest read -- one pass over them
static void batch_take_offsets(pgx_batch *b, const uint64_t *offsets, uint64_t n_reads, const char *who) {
    const uint64_t lo = offsets[0];
    HostBuf &hb = b->h_off[b->h_off_cur ^ 1]; // (swapped in once the offsets have proved valid: a refused upload leaves the batch as it was)
    hb.ensure((n_reads + 1) * 8);
    uint64_t *ho = hb.as<uint64_t>();
    ho[0] = 0;
    // ten million offsets are ~15 ms of one core: slices on a few host threads (a fresh batch per step is bound by what its host thread does
    // between the device's work: bench.py fresh_batch)
    const unsigned nt = n_reads >= (1u << 20) ? 4u : 1u;
    uint64_t longest[4] = {0, 0, 0, 0};
    bool bad[4] = {false, false, false, false};
    auto slice = [&](unsigned t) {
        const uint64_t i0 = 1 + n_reads * t / nt, i1 = 1 + n_reads * (t + 1) / nt;
        uint64_t prev = offsets[i0 - 1], mx = 0;
        bool b_ = false;
        for (uint64_t i = i0; i < i1; i++) {
            const uint64_t o = offsets[i];
            b_ |= o < prev;
            mx = std::max(mx, o - prev);
            ho[i] = o - lo;
            prev = o;
        }
        longest[t] = mx; bad[t] = b_;
    };
    if (nt == 1) slice(0);
    else {
        std::thread th[3];
        for (unsigned t = 1; t < nt; t++) th[t - 1] = std::thread(slice, t);
        slice(0);
        for (unsigned t = 1; t < nt; t++) th[t - 1].join();
    }
    uint64_t mx = 0;
    for (unsigned t = 0; t < nt; t++) {
        if (bad[t]) throw Error(PGX_ERR_ARG, std::string(who) + ": offsets must be non-decreasing");
        mx = std::max(mx, longest[t]);
    }
    if (mx >= (1ull << 31)) throw Error(PGX_ERR_UNSUPPORTED, "read longer than 2^31 bytes");
    b->h_off_cur ^= 1;
    b->n_reads = n_reads;
    b->ran = b->ran_tags = false;
    b->plan_valid = false;
    b->slot_off_valid = false;
    b->class_valid = false;
    b->ms_upload_passes = 0;
    b->max_read_len = mx;
    b->read_bytes = offsets[n_reads] - lo;
    b->offsets.ensure((n_reads + 1) * 8);
    HIPCHECK(hipMemcpyAsync(b->offsets.p, ho, (n_reads + 1) * 8, hipMemcpyHostToDevice, b->own));
}

static void batch_upload(pgx_batch *b, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads) {
    batch_take_offsets(b, offsets, n_reads, "pgx_batch_upload");
    // device offsets are rebased to 0; 32 bytes of zero padding after the last read
    // copies go through the batch's own non-blocking stream: batches of other host threads (other streams of the same device)
    // are not serialised behind them the way copies on the legacy default stream would be
    b->reads.ensure(b->read_bytes + 32);
    HIPCHECK(hipMemsetAsync((uint8_t *)b->reads.p + b->read_bytes, 0, 32, b->own));
    if (b->read_bytes) HIPCHECK(hipMemcpyAsync(b->reads.p, reads + offsets[0], b->read_bytes, hipMemcpyHostToDevice, b->own));
    HIPCHECK(hipStreamSynchronize(b->own));
}

// the reads as the host packed them (pgx_pack_reads): a quarter of the bytes over the link, and neither pgx_bad_chunks_kernel nor
// pgx_classify_reads_kernel nor their read-back on the device -- the packed words, the flags and the side list the two-step kernel wants arrive
// ready; the bytes the other kernels read are rebuilt on the device (pgx_unpack_reads_kernel + the listed reads' own bytes over them)
static void batch_upload_packed(pgx_batch *b, const uint32_t *packed, const uint64_t *offsets, uint64_t n_reads, const uint64_t *side_ids,
                                const uint8_t *side_bytes, uint64_t n_side) {
    if (offsets[0] != 0) throw Error(PGX_ERR_ARG, "pgx_batch_upload_packed: offsets[0] must be 0 (word w of the packed stream holds symbols 16 w .. 16 w + 15)");
    if (n_side > n_reads) throw Error(PGX_ERR_ARG, "pgx_batch_upload_packed: more listed reads than reads");
    for (uint64_t k = 0; k < n_side; k++) // (before anything of the batch changes: a refused upload leaves it as it was)
        if (side_ids[k] >= n_reads || (k && side_ids[k] <= side_ids[k - 1])) throw Error(PGX_ERR_ARG, "pgx_batch_upload_packed: listed read ids must ascend and lie inside the batch");
    batch_take_offsets(b, offsets, n_reads, "pgx_batch_upload_packed");
    b->h_side_off.resize(n_side + 1);
    b->h_side_off[0] = 0;
    for (uint64_t k = 0; k < n_side; k++) b->h_side_off[k + 1] = b->h_side_off[k] + (offsets[side_ids[k] + 1] - offsets[side_ids[k]]);
    const uint64_t n_chunks = (b->read_bytes + 15) >> 4, side_total = b->h_side_off[n_side];
    hipStream_t s = b->own;
    b->packed.ensure((n_chunks + 64) * 4);
    b->reads.ensure(n_chunks * 16 + 32);
    b->read_flags.ensure(((n_reads + 3) & ~3ull) + 4);
    b->side_list.ensure((n_reads ? n_reads : 1) * sizeof(pgx_heavy_item));
    b->side_count.ensure(16);
    if (n_chunks) HIPCHECK(hipMemcpyAsync(b->packed.p, packed, n_chunks * 4, hipMemcpyHostToDevice, s));
    if (n_side) {
        b->up_side_ids.ensure(n_side * 8);
        b->up_side_off.ensure((n_side + 1) * 8);
        b->up_side_bytes.ensure(side_total ? side_total : 1);
        HIPCHECK(hipMemcpyAsync(b->up_side_ids.p, side_ids, n_side * 8, hipMemcpyHostToDevice, s));
        HIPCHECK(hipMemcpyAsync(b->up_side_off.p, b->h_side_off.data(), (n_side + 1) * 8, hipMemcpyHostToDevice, s));
        if (side_total) HIPCHECK(hipMemcpyAsync(b->up_side_bytes.p, side_bytes, side_total, hipMemcpyHostToDevice, s));
    }
    for (auto &e : b->ev_up)
        if (!e) HIPCHECK(hipEventCreate(&e));
    HIPCHECK(hipEventRecord(b->ev_up[0], s));
    HIPCHECK(hipMemsetAsync(b->side_count.p, 0, 16, s));
    HIPCHECK(hipMemsetAsync(b->read_flags.p, 0, ((n_reads + 3) & ~3ull) + 4, s));
    if (n_chunks) {
        int cus = 0;
        HIPCHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device));
        hipLaunchKernelGGL(pgx_unpack_reads_kernel, dim3(std::min<unsigned>(grid_for(n_chunks, 256), (unsigned)cus * 16u)), dim3(256), 0, s, b->packed.as<uint32_t>(), n_chunks,
                           b->reads.as<uint8_t>());
        HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipMemsetAsync((uint8_t *)b->reads.p + b->read_bytes, 0, 32, s)); // (the tail of the last word unpacks to 'A's)
    if (n_side) {
        hipLaunchKernelGGL(pgx_side_reads_kernel, dim3(grid_for(n_side * 64, 256)), dim3(256), 0, s, b->reads.as<uint8_t>(), b->offsets.as<uint64_t>(), b->up_side_ids.as<uint64_t>(),
                           b->up_side_off.as<uint64_t>(), b->up_side_bytes.as<uint8_t>(), n_side, b->read_flags.as<uint8_t>(), b->side_list.as<pgx_heavy_item>(),
                           b->side_count.as<unsigned long long>());
        HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipEventRecord(b->ev_up[1], s));
    HIPCHECK(hipStreamSynchronize(s));
    HIPCHECK(hipEventElapsedTime(&b->ms_upload_passes, b->ev_up[0], b->ev_up[1]));
    b->class_valid = true; // what pgx_batch_run would otherwise find out with two passes over the bytes and a read-back
    b->class_ok = true;
    b->side_reads_est = n_side;
}

// LCE image (pgx_image.h): suffix array in text coordinates + the text at two bits per symbol, for the pairs kernel's forward stages over narrow intervals.
// Built once per device image, on the device: the suffix array by the locate kernels (every BWT run is an independent chain from its sample), the text from
// it (the first symbol of suffix i is the one whose C-bucket holds i).  Only next to a narrow PAIRS image (textbook tables, n < 2^32); PGX_FM_LCE=0: never.
static std::mutex g_lce_mutex;
static void ensure_lce(pgx_index *h, pgx_device_image *d) {
    std::lock_guard<std::mutex> lock(g_lce_mutex);
    if (d->lce_state) return;
    d->lce_state = 2;
    const char *env = std::getenv("PGX_FM_LCE");
    const uint64_t n = d->img.n;
    if ((env && env[0] == '0') || !d->img.pairs || d->img.wide || n < 4096 || n >= (1ull << 32) - (1ull << 20) || h->ri.max_length == 0) return;
    uint64_t tot[6] = {0, 0, 0, 0, 0, 0}; // symbol counts of the BWT = bucket bounds of the first column
    for (const auto &blk : h->ri.blocks)
        for (const auto &ru : blk.runs) if (ru.first < 6) tot[ru.first] += ru.second;
    const uint64_t n_seq = tot[0];
    if (tot[0] + tot[1] + tot[2] + tot[3] + tot[4] + tot[5] != n || n_seq == 0 || n_seq > (1ull << 24)) return;
    {
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) { (void)hipGetLastError(); return; }
        if ((double)mem_free < 16.0 * (double)n + (double)(2ull << 30)) return; // 8 n (suffix array as the locate kernels write it) + n (text bytes) + 5.25 n (the image) + room
    }
    DevBuf vals, seq_len, seq_start, text8, bad;
    try {
        pgx_device_image *dl = locate_image(h, d->device);
        if (!dl->loc.n || dl->loc.n != n) throw Error(PGX_ERR_UNSUPPORTED, "no locate image");
        const uint64_t first = 0, last = n - 1;
        std::vector<uint64_t> off;
        uint64_t nv = 0;
        locate_core(h, dl, &first, &last, 1, 0, off, vals, nv);
        if (nv != n) throw Error(PGX_ERR_UNSUPPORTED, "suffix array incomplete");
        const uint64_t ml = h->ri.max_length;
        seq_len.ensure(n_seq * 8); seq_start.ensure((n_seq + 1) * 8); bad.ensure(16);
        HIPCHECK(hipMemset(seq_len.p, 0, n_seq * 8));
        HIPCHECK(hipMemset(bad.p, 0, 16));
        hipLaunchKernelGGL(pgx_lce_seqlen_kernel, dim3(grid_for(n_seq, 256)), dim3(256), 0, nullptr, vals.as<uint64_t>(), n_seq, ml, seq_len.as<unsigned long long>());
        HIPCHECK(hipGetLastError());
        std::vector<uint64_t> hl(n_seq), hs(n_seq + 1, 0);
        HIPCHECK(hipMemcpy(hl.data(), seq_len.p, n_seq * 8, hipMemcpyDeviceToHost));
        for (uint64_t q = 0; q < n_seq; q++) { if (hl[q] == 0) throw Error(PGX_ERR_UNSUPPORTED, "a sequence without an endmarker suffix"); hs[q + 1] = hs[q] + hl[q]; }
        if (hs[n_seq] != n) throw Error(PGX_ERR_UNSUPPORTED, "sequence lengths do not add up to the BWT size");
        HIPCHECK(hipMemcpy(seq_start.p, hs.data(), (n_seq + 1) * 8, hipMemcpyHostToDevice));
        const uint64_t n_words = (n + 15) / 16 + 64, n_flag_words = n_words / 1024 + 2; // (64 words = two lines of padding behind the text, flagged)
        text8.ensure(n);
        d->lce_sa.ensure(n * 4 + 128); // (the kernel reads aligned windows of up to 20 entries from an interval's first entry on)
        d->lce_text.ensure(n_words * 4);
        d->lce_flags.ensure(n_flag_words * 4);
        HIPCHECK(hipMemset(d->lce_flags.p, 0, n_flag_words * 4));
        const uint64_t c1 = tot[0], c2 = c1 + tot[1], c3 = c2 + tot[2], c4 = c3 + tot[3], c5 = c4 + tot[4];
        hipLaunchKernelGGL(pgx_lce_scatter_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, nullptr, vals.as<uint64_t>(), n, ml,
                           seq_start.as<uint64_t>(), n_seq, c1, c2, c3, c4, c5, d->lce_sa.as<uint32_t>(), text8.as<uint8_t>(), bad.as<unsigned long long>());
        HIPCHECK(hipGetLastError());
        unsigned long long n_bad = 0;
        HIPCHECK(hipMemcpy(&n_bad, bad.p, 8, hipMemcpyDeviceToHost));
        if (n_bad) throw Error(PGX_ERR_UNSUPPORTED, "suffix array values outside the collection");
        vals.release();
        // both orientations of every sequence (pgx_lce_rc_check_kernel): a forward-only collection is searched stepwise, as the reference's arithmetic has it
        if (n_seq & 1) throw Error(PGX_ERR_UNSUPPORTED, "odd number of sequences");
        hipLaunchKernelGGL(pgx_lce_rc_check_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, nullptr, text8.as<uint8_t>(),
                           seq_start.as<uint64_t>(), n_seq, n, bad.as<unsigned long long>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpy(&n_bad, bad.p, 8, hipMemcpyDeviceToHost));
        if (n_bad) throw Error(PGX_ERR_UNSUPPORTED, "the collection does not hold every sequence next to its reverse complement");
        hipLaunchKernelGGL(pgx_lce_pack_kernel, dim3((unsigned)std::min<uint64_t>((n_words + 255) / 256, 1u << 20)), dim3(256), 0, nullptr, text8.as<uint8_t>(), n, n_words,
                           d->lce_text.as<uint32_t>(), d->lce_flags.as<uint32_t>());
        HIPCHECK(hipGetLastError());
        text8.release();
        const char *le = std::getenv("PGX_FM_LCP"); // (PGX_FM_LCP=0: every occurrence is compared with the text, as before the table existed)
        const bool with_lcp = !(le && le[0] == '0');
        if (with_lcp) {
            d->lce_lcp.ensure(n + 64); // (the kernel reads aligned windows of up to 20 entries)
            hipLaunchKernelGGL(pgx_lce_lcp_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, nullptr, d->lce_sa.as<uint32_t>(),
                               d->lce_text.as<uint32_t>(), d->lce_flags.as<uint32_t>(), n, d->lce_lcp.as<uint8_t>());
            HIPCHECK(hipGetLastError());
        }
        HIPCHECK(hipDeviceSynchronize());
        d->img.lce_lcp = with_lcp ? d->lce_lcp.as<uint8_t>() : nullptr;
        d->img.lce_sa = d->lce_sa.as<uint32_t>();
        d->img.lce_text = d->lce_text.as<uint32_t>();
        d->img.lce_flags = d->lce_flags.as<uint32_t>();
        d->img.lce_max = with_lcp ? PGX_LCE_MAX_OCC : 16; // (without the table of common prefixes every occurrence costs a trip)
        if (const char *e = std::getenv("PGX_FM_LCE_MAX")) d->img.lce_max = (uint32_t)std::min<unsigned long>(std::strtoul(e, nullptr, 10), (unsigned long)PGX_LCE_MAX_OCC);
        d->img.refill_min = 12; // (chr22 scale, 1 / 3 / 6 / 10 / 16 / 24: main kernel 10.76 / 10.44 / 10.24 / 10.15 / 10.10 / 10.08 ms, step 13.16 / 12.87 / 12.62 / 12.59 / 12.56 / 12.65)
        if (const char *e = std::getenv("PGX_FM_REFILL_MIN")) d->img.refill_min = (uint32_t)std::max<unsigned long>(1ul, std::min<unsigned long>(std::strtoul(e, nullptr, 10), 64ul));
        d->lce_state = 1;
    } catch (...) { // (no LCE image: the search runs on the PAIRS image alone, as before)
        (void)hipGetLastError();
        d->lce_sa.release(); d->lce_text.release(); d->lce_flags.release(); d->lce_lcp.release();
        d->img.lce_sa = nullptr; d->img.lce_text = nullptr; d->img.lce_flags = nullptr; d->img.lce_lcp = nullptr;
    }
    vals.release(); seq_len.release(); seq_start.release(); text8.release(); bad.release();
}

extern "C" pgx_status pgx_batch_create(pgx_index *h, int device, const uint8_t *reads, const uint64_t *offsets,
                                       uint64_t n_reads, pgx_batch **out) {
    PGX_GUARD_BEGIN
    if (!h || !out || !offsets || (!reads && n_reads && offsets[n_reads] != offsets[0]))
        throw Error(PGX_ERR_ARG, "pgx_batch_create: null argument");
    if (!h->has_rank) throw Error(PGX_ERR_ARG, "pgx_batch_create: index opened without an r-index");
    *out = nullptr;
    pgx_device_image *dimg = device_image(h, device);
    ensure_lce(h, dimg);
    std::unique_ptr<pgx_batch, void (*)(pgx_batch *)> b(new pgx_batch(), batch_release);
    b->h = h;
    b->dimg = dimg;
    b->device = device;
    HIPCHECK(hipStreamCreateWithFlags(&b->own, hipStreamNonBlocking));
    batch_upload(b.get(), reads, offsets, n_reads);
    *out = b.release();
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_batch_upload(pgx_batch *b, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads) {
    PGX_GUARD_BEGIN
    RoctxRange range("pgx_batch_upload");
    if (!b || !offsets || (!reads && n_reads && offsets[n_reads] != offsets[0])) throw Error(PGX_ERR_ARG, "pgx_batch_upload: null argument");
    use_device(b->device);
    batch_upload(b, reads, offsets, n_reads);
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_batch_upload_packed(pgx_batch *b, const uint32_t *packed, const uint64_t *offsets, uint64_t n_reads, const uint64_t *side_ids,
                                              const uint8_t *side_bytes, uint64_t n_side) {
    PGX_GUARD_BEGIN
    RoctxRange range("pgx_batch_upload_packed");
    if (!b || !offsets || (!packed && n_reads && offsets[n_reads] != offsets[0]) || (n_side && (!side_ids || !side_bytes)))
        throw Error(PGX_ERR_ARG, "pgx_batch_upload_packed: null argument");
    use_device(b->device);
    batch_upload_packed(b, packed, offsets, n_reads, side_ids, side_bytes, n_side);
    return PGX_OK;
    PGX_GUARD_END
}

static void record(pgx_batch *b, int i, hipStream_t s) {
    static const char *const stage[10] = {"pgx: run begins (classify, sizing)", "pgx: find_mems launches follow", "pgx: find_mems enqueued", "pgx: compaction enqueued",
                                          "pgx: tag locate enqueued", "pgx: tag gather enqueued", "pgx: tag sort/unique enqueued", "pgx: run enqueued", "pgx: main find_mems kernel enqueued",
                                          "pgx: per-upload passes enqueued"};
    if (roctx().on) roctx().mark(stage[i]);
    if (!b->timed) return;
    if (!b->ev[i]) HIPCHECK(hipEventCreate(&b->ev[i]));
    HIPCHECK(hipEventRecord(b->ev[i], s));
}

extern "C" pgx_status pgx_batch_run(pgx_batch *b, uint64_t min_len, uint64_t min_occ, uint32_t flags, void *stream) {
    PGX_GUARD_BEGIN
    RoctxRange range("pgx_batch_run");
    if (!b) throw Error(PGX_ERR_ARG, "pgx_batch_run: null batch");
    use_device(b->device);
    hipStream_t s = stream ? (hipStream_t)stream : b->own;
    PgxDevImage img = b->dimg->img; // (a copy: the seed table is chosen per run)
    // seeds need min_len >= their depth (no stage of a shorter search has room for one): the shallower table serves searches below the depth of the first
    if (img.seed_k_main && min_len < img.seed_k_main && img.seed_k_small && min_len >= img.seed_k_small) { img.seed = img.seed_small; img.seed_k = img.seed_k_small; }
    const uint64_t n = b->n_reads;
    const bool want_tags = (flags & PGX_RUN_TAGS) != 0;
    if (want_tags && !b->h->has_tags) throw Error(PGX_ERR_ARG, "pgx_batch_run: PGX_RUN_TAGS without a tag array");
    b->timed = (flags & PGX_RUN_TIMING) != 0;
    b->ran = false;
    b->ran_tags = false;
    b->n_mems = b->n_positions = b->n_ext = b->n_tag_overflow = 0;
    std::memset(&b->timing, 0, sizeof b->timing);

    // Speculative sizing: a run normally reads a few scalars back in mid-flight (MEM total, tag-stage totals) because they size
    // the next buffers -- each a host synchronisation with the device idle meanwhile.  When the previous run of this batch had the
    // same shape (reads, min_len, min_occ, tags), the buffers and grids are sized from ITS totals (+ 25 %), all counts stay on the
    // device, capacity checks raise an abort flag there, and the host reads everything once at the end; if the flag came up (or the
    // 32-bit state overflowed) the run is repeated in exact mode.  PGX_SPEC=0 switches it off.
    bool force_worst = false; // the arena of a speculative pass overflowed: the exact pass uses the worst-case slot layout
    for (int pass = 0;; pass++) {
    b->n_mems = b->n_positions = b->n_ext = b->n_tag_overflow = 0;
    std::memset(&b->timing, 0, sizeof b->timing);
    b->counters.ensure(PGX_CTR_ALL * 8); // layout: PgxCounterSlot (pgx_device.h)
    HIPCHECK(hipMemsetAsync(b->counters.p, 0, PGX_CTR_ALL * 8, s));
    unsigned long long *d_next = b->counters.as<unsigned long long>();
    unsigned long long *d_nover = d_next + PGX_CTR_TAG_OVERFLOW;

    record(b, 0, s);
    // 1. worst-case MEM slots per read: cap = min(len, len - min_len + 1).  The slot buffer is bounded by
    //    a budget; batches whose worst case exceeds it are processed in chunks of consecutive reads.
    b->slot_off.ensure((n + 1) * 8);
    bool fresh_work = false, fresh_mark = false; // this run performs passes only the first run after an upload needs (pgx_timing.ms_per_upload); event 9 recorded behind them
    if (!b->slot_off_valid || b->slot_off_min_len != min_len) { // depends on the reads and min_len only: kept across runs
        fresh_work = true;
        scan_excl(2, b->offsets.p, n, min_len, b->slot_off.as<uint64_t>(), b->scan_tmp, s);
        b->slot_off_valid = true;
        b->slot_off_min_len = min_len;
    }
    b->mem_count.ensure((n ? n : 1) * 4);
    b->mem_off.ensure((n + 1) * 8);
    // a quarter of the device's memory (72 GB of the MI355X's 288 GB: ten million 150-bp reads are one chunk), 16 GiB at least
    uint64_t budget_slots = (16ull << 30) / sizeof(pgx_mem);
    {
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) budget_slots = std::max<uint64_t>(budget_slots, (uint64_t)(mem_total / 4) / sizeof(pgx_mem));
        else (void)hipGetLastError();
    }
    if (const char *e = std::getenv("PGX_SLOT_BUDGET_MB")) budget_slots = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10)) * (1ull << 20) / sizeof(pgx_mem);
    if (!b->plan_valid || b->plan_min_len != min_len || b->plan_budget != budget_slots) { // cached across runs
        b->chunks.clear();
        uint64_t r0 = 0, base = 0, acc = 0;
        // the slots of a read never exceed its length: a batch whose bytes fit the budget is one chunk, no per-read loop
        if (n && b->read_bytes <= budget_slots) b->chunks.push_back({0, n, 0, std::max<uint64_t>(b->read_bytes, 1)});
        else {
            for (uint64_t i = 0; i < n; i++) {
                const uint64_t len = b->h_offsets()[i + 1] - b->h_offsets()[i];
                const uint64_t cap = len < min_len ? 0 : std::min<uint64_t>(len, len - min_len + 1);
                if (acc && acc + cap > budget_slots) { b->chunks.push_back({r0, i, base, acc}); r0 = i; base += acc; acc = 0; }
                acc += cap;
            }
            if (n) b->chunks.push_back({r0, n, base, acc});
        }
        b->plan_valid = true; b->plan_min_len = min_len; b->plan_budget = budget_slots;
    }
    const std::vector<pgx_chunk> &chunks = b->chunks;
    uint64_t max_slots = 1;
    for (auto &c : chunks) max_slots = std::max(max_slots, c.slots);
    uint64_t max_chunk_reads = 1;
    for (auto &c : chunks) max_chunk_reads = std::max(max_chunk_reads, c.r1 - c.r0);
    // The slot buffer: a dense array of the first four MEMs of every read (PGX_FAST_SLOTS, pgx_kernels.hip pgx_slot_index) + either an ARENA for the
    // fifth and later MEMs, sized from the last run of this batch (or two slots per read), or -- PGX_SLOT_ARENA=0, tiny batches, and the repeat of a chunk
    // whose arena proved too small -- the worst-case region (max_slots).  Sized per chunk below.
    const char *arena_env = std::getenv("PGX_SLOT_ARENA");
    const bool arena_on = !(arena_env && arena_env[0] == '0') && !force_worst;
    b->ovf_base.ensure((max_chunk_reads ? max_chunk_reads : 1) * 4);
    // 2. the hot kernel, 3. CSR offsets + compaction (per chunk)
    float ms_fm = 0, ms_cp = 0, ms_main = 0;
    uint64_t mem_base = 0;
    int occ = 0, cus = 0;
    const void *kfn = nullptr, *kfn_wide = nullptr, *kfn_pairs = nullptr;
    size_t pairs_lds = 0;
    uint64_t n_ext_host = 0;
    if (n) {
        // persistent grid: as many workgroups as the device keeps resident (no inter-workgroup
        // dependency exists, so any grid size is correct; this one avoids a tail of late blocks)
        const bool in_lds = b->dimg->lds_bytes != 0, dense = img.dense != 0, d2 = img.dense == 2, d3 = img.dense == 3;
        const bool seeded = img.seed_k != 0 && min_len >= img.seed_k; // (no stage of a shorter search has room for a seed)
        kfn_wide = in_lds ? (dense ? (seeded ? (const void *)pgx_find_mems_kernel<true, 1, false, true> : (const void *)pgx_find_mems_kernel<true, 1, false, false>)
                                   : (const void *)pgx_find_mems_kernel<true, 0, false, false>)
                   : d3   ? (seeded ? (const void *)pgx_find_mems_kernel<false, 3, false, true> : (const void *)pgx_find_mems_kernel<false, 3, false, false>)
                   : d2   ? (seeded ? (const void *)pgx_find_mems_kernel<false, 2, false, true> : (const void *)pgx_find_mems_kernel<false, 2, false, false>)
                          : (dense ? (seeded ? (const void *)pgx_find_mems_kernel<false, 1, false, true> : (const void *)pgx_find_mems_kernel<false, 1, false, false>)
                                   : (const void *)pgx_find_mems_kernel<false, 0, false, false>);
        kfn = kfn_wide;
        // 32-bit interval state for dense images of BWTs shorter than 2^30 (PGX_FM_NARROW=0 switches it off)
        const char *nv = std::getenv("PGX_FM_NARROW");
        bool c_fits = true; // C[] comes straight from the file: a (corrupt) value beyond 2^32 must not be truncated by the 32-bit state
        for (int i = 0; i < 8; i++) c_fits = c_fits && !(b->h->img.consts.C[i] >> 32);
        if (dense && !d3 && img.n < (1ull << 30) && c_fits && !(nv && nv[0] == '0'))
            kfn = in_lds ? (seeded ? (const void *)pgx_find_mems_kernel<true, 1, true, true> : (const void *)pgx_find_mems_kernel<true, 1, true, false>)
                  : d2   ? (seeded ? (const void *)pgx_find_mems_kernel<false, 2, true, true> : (const void *)pgx_find_mems_kernel<false, 2, true, false>)
                         : (seeded ? (const void *)pgx_find_mems_kernel<false, 1, true, true> : (const void *)pgx_find_mems_kernel<false, 1, true, false>);
        // two extensions per cache line where the index has a PAIRS image (PGX_FM_PAIRS=0: the dense2 kernel alone); the kernel chosen
        // above then serves the reads the pairs kernel skips (a byte outside A C G T), on the second stream
        const char *pv = std::getenv("PGX_FM_PAIRS");
        // (only behind the seed table: the wide intervals at the start of an unseeded stage always have special positions between their ends)
        if (img.pairs && seeded && !(n >> 32) && b->read_bytes < (1ull << 35) && !(pv && pv[0] == '0'))
        {
            const bool s64 = img.pairs_stride == PGX_PAIRS_STRIDE64;
            kfn_pairs = img.wide ? (s64 ? (const void *)pgx_find_mems_pairs_kernel<true, true, false, false, true, false> : (const void *)pgx_find_mems_pairs_kernel<true, true, false, false, false, false>)
                                 : (s64 ? (const void *)pgx_find_mems_pairs_kernel<true, false, false, false, true, false> : (const void *)pgx_find_mems_pairs_kernel<true, false, false, false, false, false>);
        }
        pairs_lds = img.wide ? (size_t)img.n_sbp * 192 : 0; // (superblock bases of the wide form, behind the other dynamic LDS)
        HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn_pairs ? kfn_pairs : kfn, PGX_FM_THREADS, kfn_pairs ? pairs_lds : b->dimg->lds_bytes));
        HIPCHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device));
        if (occ < 1) occ = 1;
        if (const char *e = std::getenv("PGX_FM_WG_PER_CU")) { // experiments: fewer resident workgroups per CU
            const int v = std::atoi(e);
            if (v >= 1 && v < occ) occ = v;
        }
    }
    uint32_t heavy_ext = PGX_FM_HEAVY_EXT; // extensions on one read before its rest goes to the heavy-read kernel (0 = never)
    if (const char *e = std::getenv("PGX_FM_HEAVY_EXT")) heavy_ext = (uint32_t)std::strtoul(e, nullptr, 10);
    unsigned long long *d_heavy_count = d_next + PGX_CTR_HEAVY;
    if (heavy_ext) {
        b->heavy_list.ensure((size_t)PGX_FM_HEAVY_CAP * sizeof(pgx_heavy_item));
        b->heavy_scratch.ensure((size_t)PGX_FM_HEAVY_GRID * PGX_FM_HEAVY_MAXLEN * sizeof(PgxHeavyResult));
    }
    unsigned long long *d_cursor = d_next + PGX_CTR_CURSOR;
    const char *spec_env = std::getenv("PGX_SPEC");
    const bool spec = pass == 0 && chunks.size() == 1 && b->shape_valid && b->shape_reads == n && b->shape_min_len == min_len && b->shape_min_occ == min_occ &&
                      b->shape_tags == want_tags && (!want_tags || (b->tw.have_last && b->tw.last_largest <= PGX_SORT_WG_LDS_CAP)) &&
                      !(spec_env && spec_env[0] == '0') && !std::getenv("PGX_FM_NARROW_FORCE_REDO");
    const uint64_t cm_cap = with_slack(b->last_mems);
    uint64_t *d_abort = reinterpret_cast<uint64_t *>(d_next + PGX_CTR_ABORT);
    for (size_t ci = 0; ci < chunks.size(); ci++) {
        const pgx_chunk &c = chunks[ci];
        const uint64_t cn = c.r1 - c.r0;
        record(b, 1, s);
        unsigned grid = grid_for(cn, 64);
        int wg = occ;
        if (img.dense && !std::getenv("PGX_FM_WG_PER_CU")) {
            // the dense kernels need little occupancy, and every resident lane ends the launch inside a read (the tail):
            // aim at >= 5 reads per lane (1 M reads: 3 workgroups per CU measured best, 493 vs 477 (x) and 179 vs 160 (synth) Mreads/s)
            const uint64_t want = cn / (5ull * (uint64_t)cus * PGX_FM_THREADS);
            wg = (int)std::min<uint64_t>((uint64_t)occ, std::max<uint64_t>(2, want));
        }
        if (grid > (unsigned)(wg * cus)) grid = (unsigned)(wg * cus);
        uint64_t *local = b->mem_off.as<uint64_t>() + c.r0; // local CSR offsets of this chunk (scratch until the global scan below)
        uint64_t cm = 0;
        const void *kf = kfn;
        // arena for the fifth and later MEMs of this chunk's reads (0 = worst-case layout)
        uint64_t ovf_cap = 0;
        if (arena_on) {
            uint64_t want = b->shape_valid && b->shape_reads == n && b->shape_min_len == min_len && chunks.size() == 1 ? with_slack(b->last_ovf_used) + 4096 : 8 * cn + 4096; // (first run of a shape: eight slots per read; chr22 scale asks for 2.6, the x fixture for 6.7)
            if (const char *e = std::getenv("PGX_SLOT_ARENA_CAP")) want = std::strtoull(e, nullptr, 10); // tests: an arena that overflows
            want = std::max<uint64_t>(want, b->max_read_len + 1); // (an overflowing extent is parked at the start of the arena: it must fit)
            want = std::max<uint64_t>(want, (uint64_t)PGX_ARENA_SUBS * (b->max_read_len + 1)); // (every sub-arena must hold a parked extent)
            want = (want + PGX_ARENA_SUBS - 1) / PGX_ARENA_SUBS * PGX_ARENA_SUBS;
            if (want < c.slots && want < (1ull << 32)) ovf_cap = want; // otherwise the worst case is no bigger
        }
        bool arena_failed = false;
        for (int attempt = 0;; attempt++) {
            if (arena_failed) ovf_cap = 0;
            b->slots.ensure(((ovf_cap ? ovf_cap : c.slots) + 4 * cn) * sizeof(pgx_mem));
            // per-chunk counters (slots below PGX_CTR_TAG0): extensions, cursors, heavy reads, 32-bit overflow flag, MEMs of the chunk
            if (ci || attempt) {
                HIPCHECK(hipMemsetAsync(d_next, 0, PGX_CTR_TAG0 * 8, s));
                HIPCHECK(hipMemsetAsync(d_next + PGX_CTR_ARENA0, 0, (PGX_CTR_ALL - PGX_CTR_ARENA0) * 8, s));
            }
            const uint8_t *a_reads = b->reads.as<uint8_t>();
            const uint64_t *a_off = b->offsets.as<uint64_t>(), *a_slot_off = b->slot_off.as<uint64_t>();
            uint64_t a_n = c.r1, a_min_len = min_len, a_min_occ = min_occ, a_base = c.slot_base, a_first = c.r0;
            pgx_mem *a_slots = b->slots.as<pgx_mem>();
            uint32_t *a_cnt = b->mem_count.as<uint32_t>();
            unsigned long long *a_next = d_next, *a_cur = d_cursor;
            PgxDevImage a_img = img;
            uint32_t a_hext = heavy_ext, a_hcap = PGX_FM_HEAVY_CAP;
            uint32_t *a_ovf = b->ovf_base.as<uint32_t>() - c.r0; // (indexed by read id; the buffer holds this chunk's reads)
            uint64_t a_ovf_cap = ovf_cap;
            pgx_heavy_item *a_hlist = b->heavy_list.as<pgx_heavy_item>();
            unsigned long long *a_hcount = d_heavy_count;
            const pgx_heavy_item *a_rlist = nullptr;
            const unsigned long long *a_rcount = nullptr;
            bool side_running = false;
            if (kfn_pairs) { // the pairs kernel; the kernel chosen above serves the reads it skips
                // reads with a byte outside A C G T cannot be seeded: they go to the dense2 kernel at once, on a second stream next to the pairs
                // kernel, which skips them (a read cut from an N run is a chain of thousands of extensions: behind the pairs kernel it was 1.5 ms of tail)
                const uint8_t *a_skip = nullptr;
                if (chunks.size() == 1 && !std::getenv("PGX_FM_NO_SIDE")) {
                    if (!b->side) {
                        HIPCHECK(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
                        for (auto &e : b->ev_side) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    }
                    if (!b->class_valid) { // once per upload: two small passes and one scalar read back
                        fresh_work = true;
                        const uint64_t cap = std::max<uint64_t>(cn, 1024); // chunks with such a byte; beyond that (lower-case reads, say) no side launch
                        b->read_flags.ensure(((cn + 3) & ~3ull) + 4);
                        b->side_list.ensure((cn ? cn : 1) * sizeof(pgx_heavy_item));
                        b->side_count.ensure(16);
                        b->scan_tmp.ensure(cap * 8);
                        b->scan_tmp.scan_epoch = 0; // (the list overwrites the scans' tile words: the next scan clears the buffer)
                        b->packed.ensure(((b->read_bytes + 15) / 16 + 64) * 4); // the reads as two bits per symbol (written by the same pass)
                        HIPCHECK(hipMemsetAsync(b->side_count.p, 0, 16, s));
                        HIPCHECK(hipMemsetAsync(b->read_flags.p, 0, ((cn + 3) & ~3ull) + 4, s));
                        unsigned long long *d_bad = b->side_count.as<unsigned long long>() + 1;
                        hipLaunchKernelGGL(pgx_bad_chunks_kernel, dim3(std::min<unsigned>(grid_for((b->read_bytes + 15) / 16, 256), (unsigned)cus * 16u)), dim3(256), 0, s,
                                           a_reads, b->read_bytes, b->scan_tmp.as<uint64_t>(), d_bad, cap, b->packed.as<uint32_t>());
                        HIPCHECK(hipGetLastError());
                        unsigned long long n_bad = 0;
                        read_scalars(&n_bad, d_bad, sizeof n_bad, s);
                        b->class_ok = n_bad <= cap;
                        b->side_reads_est = n_bad / 4; // (a read that overlaps an N run holds a handful of such 16-byte chunks)
                        if (b->class_ok && n_bad) {
                            hipLaunchKernelGGL(pgx_classify_reads_kernel, dim3(grid_for(n_bad, 256)), dim3(256), 0, s, a_reads, a_off, cn, (const uint64_t *)b->scan_tmp.as<uint64_t>(),
                                               (const unsigned long long *)d_bad, cap, b->read_flags.as<uint32_t>(), b->side_list.as<pgx_heavy_item>(),
                                               b->side_count.as<unsigned long long>());
                            HIPCHECK(hipGetLastError());
                        }
                        b->class_valid = true;
                    }
                  record(b, 9, s); fresh_mark = true;
                  if (b->class_ok) {
                    a_skip = b->read_flags.as<uint8_t>();
                    const char *sse = std::getenv("PGX_FM_SIDE_SERIAL"); // (experiment: the launch in front of the pairs kernel on the same stream, not next to it)
                    hipStream_t side_stream = (sse && sse[0] == '1') ? s : b->side;
                    HIPCHECK(hipEventRecord(b->ev_side[0], s));
                    HIPCHECK(hipStreamWaitEvent(side_stream, b->ev_side[0], 0));
                    const pgx_heavy_item *s_list = b->side_list.as<pgx_heavy_item>();
                    const unsigned long long *s_count = b->side_count.as<unsigned long long>();
                    unsigned long long *s_cur = d_next + PGX_CTR_SIDE_CURSOR;
                    // (a lane of this launch walks its read alone, one dependent extension after the other next to the pairs kernel: the launch lasts as long
                    //  as its longest chain, so its reads go to the heavy-read kernel -- every start position at once -- earlier than the main launch's)
                    uint32_t s_hext = heavy_ext ? std::min<uint32_t>(heavy_ext, PGX_FM_SIDE_HEAVY_EXT) : 0u;
                    if (const char *e = std::getenv("PGX_FM_SIDE_HEAVY_EXT")) s_hext = (uint32_t)std::strtoul(e, nullptr, 10);
                    void *sargs[] = {&a_img, &a_reads, &a_off, &a_n, &a_min_len, &a_min_occ, &a_slot_off, &a_slots, &a_cnt, &a_next, &s_cur, &a_first, &a_base,
                                     &s_hext, &a_hcap, &a_hlist, &a_hcount, &s_list, &s_count, &a_ovf, &a_ovf_cap};
                    // one workgroup per CU next to the pairs kernel while these reads are few (0.4 % of the chr22 workload: 43 k reads, less than one per lane);
                    // with many of them the launch was the longest thing in the step (5 % = 500 k reads, 7.6 per lane one after the other: 21.8 ms next
                    // to a 17 ms pairs kernel): up to four per CU, two reads per lane
                    uint64_t side_min = (uint64_t)cus, side_per_lane = 2;
                    if (const char *e = std::getenv("PGX_FM_SIDE_WGS_MIN")) side_min = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
                    if (const char *e = std::getenv("PGX_FM_SIDE_PER_LANE")) side_per_lane = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
                    const unsigned side_wgs = (unsigned)std::min<uint64_t>(4ull * (uint64_t)cus, std::max<uint64_t>(side_min, b->side_reads_est / (side_per_lane * PGX_FM_THREADS) + 1));
                    HIPCHECK(hipLaunchKernel(kf, dim3(std::min<unsigned>(grid, side_wgs)), dim3(PGX_FM_THREADS), sargs, b->dimg->lds_bytes, side_stream));
                    HIPCHECK(hipEventRecord(b->ev_side[1], side_stream));
                    side_running = true;
                  }
                }
                // the reads from LDS, two bits per symbol, when every read the launch serves is pure A C G T (the others are skipped) and a
                // thread's column stays small enough for four workgroups per CU (reads up to ~350 bp); PGX_FM_PACKED=0 switches it off
                const void *kp = kfn_pairs;
                size_t plds = pairs_lds;
                const uint32_t *a_packed = nullptr;
                uint32_t a_pkw = 0;
                unsigned pgrid = grid;
                {
                    const uint32_t pkw = (uint32_t)((15 + b->max_read_len + 15) >> 4) + 1u; // words of the longest read at the worst phase + one of padding
                    const char *pe = std::getenv("PGX_FM_PACKED");
                    if (a_skip && pkw <= 24 && !(pe && pe[0] == '0')) {
                        // cooperative line fetches (one address translation per line instead of five) for PAIRS images beyond the reach of the
                        // translation caches, ~3 GB (profiles/r03_ubench_gather_loads_per_line.txt); PGX_FM_COOP=0 / 1 overrides
                        const bool s64 = img.pairs_stride == PGX_PAIRS_STRIDE64;
                        size_t lce_lds = 0;
                        bool coop = b->h->img.pairs.size() > (3ull << 30);
                        if (const char *ce = std::getenv("PGX_FM_COOP")) coop = ce[0] == '1';
#define PGX_PK(W, C) (s64 ? (const void *)pgx_find_mems_pairs_kernel<true, W, true, C, true, false> : (const void *)pgx_find_mems_pairs_kernel<true, W, true, C, false, false>)
                        kp = coop ? (img.wide ? PGX_PK(true, true) : PGX_PK(false, true)) : (img.wide ? PGX_PK(true, false) : PGX_PK(false, false));
#undef PGX_PK
                        // forward stages over narrow intervals through the suffix array and the text (pgx_image.h "LCE image"; min_occ <= 1: the longest match decides)
                        const char *le = std::getenv("PGX_FM_LCE");
                        if (img.lce_sa && !coop && !img.wide && min_occ <= 1 && !(le && le[0] == '0')) {
                            kp = s64 ? (const void *)pgx_find_mems_pairs_kernel<true, false, true, false, true, true> : (const void *)pgx_find_mems_pairs_kernel<true, false, true, false, false, true>;
                            b->timing.pairs_reads = 4u;
                            lce_lds = (size_t)PGX_FM_THREADS * (16 + 4 + 20); // per thread: a seed entry, a suffix array entry, sixteen common prefixes from any byte on (five dwords)
                        }
                        a_packed = b->packed.as<uint32_t>();
                        a_pkw = pkw;
                        plds = (size_t)pkw * PGX_FM_THREADS * 4 + (coop ? (size_t)(PGX_FM_THREADS / 64) * 8192 : 0) + (img.wide ? (size_t)img.n_sbp * 192 : 0) + lce_lds;
                        if (b->timing.pairs_reads != 4u) b->timing.pairs_reads = coop ? 3u : 2u;
                        int occ_p = 0;
                        HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_p, kp, PGX_FM_THREADS, plds));
                        if (occ_p < 1) occ_p = 1;
                        if (const char *e = std::getenv("PGX_FM_WG_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v < occ_p) occ_p = v; }
                        else occ_p = (int)std::min<uint64_t>((uint64_t)occ_p, std::max<uint64_t>(2, cn / (5ull * (uint64_t)cus * PGX_FM_THREADS)));
                        pgrid = std::min(grid_for(cn, 64), (unsigned)(occ_p * cus));
                    }
                }
                void *pargs[] = {&a_img, &a_reads, &a_off, &a_n, &a_min_len, &a_min_occ, &a_slot_off, &a_slots, &a_cnt, &a_next, &a_cur, &a_first, &a_base,
                                 &a_hext, &a_hcap, &a_hlist, &a_hcount, &a_skip, &a_packed, &a_pkw, &a_ovf, &a_ovf_cap};
                HIPCHECK(hipLaunchKernel(kp, dim3(pgrid), dim3(PGX_FM_THREADS), pargs, plds, s));
                record(b, 8, s);
            } else { // (the pairs kernel serves every read of the launch itself: where its image cannot answer, it takes that extension through the other one)
                void *args[] = {&a_img, &a_reads, &a_off, &a_n, &a_min_len, &a_min_occ, &a_slot_off, &a_slots, &a_cnt, &a_next, &a_cur, &a_first, &a_base,
                                &a_hext, &a_hcap, &a_hlist, &a_hcount, &a_rlist, &a_rcount, &a_ovf, &a_ovf_cap};
                if (ci == 0 && attempt == 0) { record(b, 9, s); fresh_mark = true; }
                HIPCHECK(hipLaunchKernel(kf, dim3(grid), dim3(PGX_FM_THREADS), args, b->dimg->lds_bytes, s)); // one of the variants
                record(b, 8, s);
            }
            if (side_running) HIPCHECK(hipStreamWaitEvent(s, b->ev_side[1], 0)); // the other stream's reads are done (they may have queued heavy reads)
            if (heavy_ext) { // the rest of reads that spent heavy_ext extensions (usually none: the launch then costs a few microseconds)
                if (b->dimg->lds_bytes)
                    hipLaunchKernelGGL(pgx_find_mems_heavy_kernel<true>, dim3(PGX_FM_HEAVY_GRID), dim3(256), b->dimg->lds_bytes, s, img, a_reads, a_off,
                                       min_len, min_occ, a_slot_off, c.slot_base, a_slots, a_cnt, d_next, (const pgx_heavy_item *)a_hlist,
                                       (const unsigned long long *)d_heavy_count, (uint32_t)PGX_FM_HEAVY_CAP, b->heavy_scratch.as<PgxHeavyResult>(), c.r0, cn, a_ovf, a_ovf_cap);
                else
                    hipLaunchKernelGGL(pgx_find_mems_heavy_kernel<false>, dim3(PGX_FM_HEAVY_GRID), dim3(256), 0, s, img, a_reads, a_off, min_len, min_occ,
                                       a_slot_off, c.slot_base, a_slots, a_cnt, d_next, (const pgx_heavy_item *)a_hlist,
                                       (const unsigned long long *)d_heavy_count, (uint32_t)PGX_FM_HEAVY_CAP, b->heavy_scratch.as<PgxHeavyResult>(), c.r0, cn, a_ovf, a_ovf_cap);
            }
            if (ovf_cap) hipLaunchKernelGGL(pgx_arena_demand_kernel, dim3(1), dim3(PGX_ARENA_SUBS), 0, s, d_next);
            HIPCHECK(hipGetLastError());
            b->timing.find_mems_launches++;
            record(b, 2, s);
            scan_excl(0, b->mem_count.as<uint32_t>() + c.r0, cn, 0, local, b->scan_tmp, s, reinterpret_cast<uint64_t *>(d_next + PGX_CTR_MEMS));
            if (spec) { cm = cm_cap; break; } // nothing is read back: the MEM total stays on the device
            unsigned long long cc[16];
            read_scalars(cc, d_next, sizeof cc, s);
            if (ovf_cap) b->last_ovf_used = cc[PGX_CTR_OVF_TOP]; // (what the reads asked for, whether or not it fitted: sizes the next arena)
            if (ovf_cap && cc[PGX_CTR_OVF_ABORT]) { arena_failed = true; continue; } // the arena was too small: once more in the worst-case layout
            const bool forced = attempt == 0 && kf != kfn_wide && std::getenv("PGX_FM_NARROW_FORCE_REDO") != nullptr; // tests
            if ((cc[PGX_CTR_OVF32] || forced) && kf != kfn_wide) { kf = kfn_wide; continue; } // a coordinate left 32 bits: repeat the chunk in 64 bits
            n_ext_host += cc[PGX_CTR_EXT];
            b->timing.heavy_reads += (uint32_t)std::min<unsigned long long>(cc[PGX_CTR_HEAVY], PGX_FM_HEAVY_CAP);
            b->timing.pairs_other_steps += (uint32_t)cc[PGX_CTR_REDO];
            cm = cc[PGX_CTR_MEMS];
            break;
        }
        b->mems.ensure_keep((mem_base + cm ? mem_base + cm : 1) * sizeof(pgx_mem), mem_base * sizeof(pgx_mem));
        hipLaunchKernelGGL(pgx_compact_mems_kernel, dim3(grid_for(cn, 256)), dim3(256), 0, s, c.r0, cn, b->slot_off.as<uint64_t>(),
                           c.slot_base, b->slots.as<pgx_mem>(), b->mem_count.as<uint32_t>(), (const uint64_t *)local, mem_base,
                           b->mems.as<pgx_mem>(), spec ? cm_cap : ~0ull, d_abort, (const uint32_t *)(b->ovf_base.as<uint32_t>() - c.r0), ovf_cap);
        HIPCHECK(hipGetLastError());
        record(b, 3, s);
        mem_base += cm;
        if (b->timed && chunks.size() > 1) { // events are reused per chunk: accumulate now
            HIPCHECK(hipStreamSynchronize(s));
            float t1 = 0, t2 = 0, t3 = 0;
            HIPCHECK(hipEventElapsedTime(&t1, b->ev[1], b->ev[2]));
            HIPCHECK(hipEventElapsedTime(&t2, b->ev[2], b->ev[3]));
            HIPCHECK(hipEventElapsedTime(&t3, b->ev[1], b->ev[8]));
            ms_fm += t1; ms_cp += t2; ms_main += t3;
        }
    }
    b->n_mems = mem_base;
    if (!kfn_pairs) b->timing.pairs_reads = 0u;
    else if (!b->timing.pairs_reads) b->timing.pairs_reads = 1u; // (2 / 3 when the launch used the packed reads / the cooperative fetches too)
    b->timing.seed_depth = (img.seed_k != 0 && min_len >= img.seed_k && img.dense) ? img.seed_k : 0u;
    if (chunks.size() != 1) { // global CSR offsets (a single chunk's local offsets already are global)
        if (chunks.empty()) { record(b, 1, s); record(b, 2, s); }
        scan_excl(0, b->mem_count.p, n, 0, b->mem_off.as<uint64_t>(), b->scan_tmp, s);
        b->mems.ensure_keep((b->n_mems ? b->n_mems : 1) * sizeof(pgx_mem), b->n_mems * sizeof(pgx_mem));
        record(b, 3, s);
    }
    record(b, 3, s);
    // 4. tag queries (find_mems.cpp:129)
    if (want_tags) {
        unsigned long long *d_nbig = d_next + PGX_CTR_TAG0;
        tag_pipeline(img, b->mems.as<pgx_mem>(), nullptr, nullptr, b->n_mems, b->tw, d_nover, d_nbig, s,
                     [&](int stage) { record(b, 4 + stage, s); }, spec, reinterpret_cast<const uint64_t *>(d_next + PGX_CTR_MEMS), d_abort);
        b->n_positions = b->tw.n_positions;
        b->ran_tags = true;
    }
    record(b, 7, s);
    unsigned long long cnt[PGX_CTR_SLOTS];
    read_scalars(cnt, b->counters.p, sizeof cnt, s);
    if (spec) {
        b->spec_runs++;
        if (cnt[PGX_CTR_OVF_ABORT]) { force_worst = true; b->last_ovf_used = cnt[PGX_CTR_OVF_TOP]; } // (the arena sized from the last run overflowed: the next one is sized from this demand)
        if (std::getenv("PGX_DEBUG_COUNTERS"))
            std::fprintf(stderr, "[pgx] speculative run: abort flags %llu, 32-bit overflow %llu, arena overflow %llu (top %llu), MEMs %llu of capacity %llu\n", cnt[PGX_CTR_ABORT], cnt[PGX_CTR_OVF32],
                         cnt[PGX_CTR_OVF_ABORT], cnt[PGX_CTR_OVF_TOP], cnt[PGX_CTR_MEMS], (unsigned long long)cm_cap);
        if (cnt[PGX_CTR_ABORT] || cnt[PGX_CTR_OVF32] || cnt[PGX_CTR_OVF_ABORT] || cnt[PGX_CTR_MEMS] > cm_cap) { b->spec_fallbacks++; b->ran_tags = false; continue; } // a capacity was too small: once more, exactly
        b->last_ovf_used = cnt[PGX_CTR_OVF_TOP];
        b->n_mems = cnt[PGX_CTR_MEMS];
        n_ext_host = cnt[PGX_CTR_EXT];
        b->timing.heavy_reads = (uint32_t)std::min<unsigned long long>(cnt[PGX_CTR_HEAVY], PGX_FM_HEAVY_CAP);
        b->timing.pairs_other_steps = (uint32_t)cnt[PGX_CTR_REDO];
        if (want_tags) {
            TagWork &w = b->tw;
            const unsigned long long *tc = cnt + PGX_CTR_TAG0; // (scalars of tag_pipeline)
            w.last_big = tc[0]; w.last_large = tc[1]; w.last_largest = tc[2]; w.last_G = tc[3]; w.last_small = tc[5];
            w.last_rep = tc[6]; w.last_dup = tc[7]; w.last_P = tc[8];
            w.n_positions = tc[8];
            b->n_positions = tc[8];
        }
    }
    // the kernels' own traffic counters (accumulated over the chunks of the run)
    if (kfn_pairs) {
        b->timing.main_lines = cnt[PGX_CTR_PAIRS_LINES]; b->timing.main_seed_loads = cnt[PGX_CTR_PAIRS_SEEDS];
        b->timing.other_lines = cnt[PGX_CTR_FM_LINES]; b->timing.other_seed_loads = cnt[PGX_CTR_FM_SEEDS];
        b->timing.two_step_trips = cnt[PGX_CTR_PAIRS_TWO];
    } else { b->timing.main_lines = cnt[PGX_CTR_FM_LINES]; b->timing.main_seed_loads = cnt[PGX_CTR_FM_SEEDS]; }
    b->last_mems = b->n_mems;
    b->shape_valid = true; b->shape_reads = n; b->shape_min_len = min_len; b->shape_min_occ = min_occ; b->shape_tags = want_tags;
    if (cnt[PGX_CTR_ST_TRIPS] && std::getenv("PGX_FM_STATS")) // only a -DPGX_FM_STATS build of the kernels fills these (scripts/fm_stats.sh)
        std::fprintf(stderr, "[pgx] find_mems wave trips %llu, live lane-trips %llu (%.1f%% of lanes), longest wave %llu trips, extensions %llu\n", cnt[PGX_CTR_ST_TRIPS],
                     cnt[PGX_CTR_ST_LIVE], 100.0 * (double)cnt[PGX_CTR_ST_LIVE] / (64.0 * (double)cnt[PGX_CTR_ST_TRIPS]), cnt[PGX_CTR_ST_LONGEST], cnt[PGX_CTR_EXT]);
    if (cnt[PGX_CTR_ST_PAIR_TRIPS] && std::getenv("PGX_FM_STATS"))
        std::fprintf(stderr, "[pgx] pairs kernel wave trips %llu, live lane-trips %llu (%.1f%%), with two extensions %llu, waiting for a second block %llu, fresh %llu, extensions through the other image %llu\n",
                     cnt[PGX_CTR_ST_PAIR_TRIPS], cnt[PGX_CTR_ST_PAIR_LIVE], 100.0 * (double)cnt[PGX_CTR_ST_PAIR_LIVE] / (64.0 * (double)cnt[PGX_CTR_ST_PAIR_TRIPS]),
                     cnt[PGX_CTR_PAIRS_TWO], cnt[PGX_CTR_ST_PAIR_WAIT], cnt[PGX_CTR_ST_PAIR_FRESH], cnt[PGX_CTR_REDO]);
    if (cnt[PGX_CTR_ST_PAIR_T_TOTAL] && std::getenv("PGX_FM_STATS"))
        std::fprintf(stderr, "[pgx] pairs kernel clock ticks: %.1f%% of the waves' time in the refill loop (%llu of %llu), %llu trips with a refill round; waiting for the seed entry %.1f%%, then for the block line %.1f%%\n",
                     100.0 * (double)cnt[PGX_CTR_ST_PAIR_T_REFILL] / (double)cnt[PGX_CTR_ST_PAIR_T_TOTAL], cnt[PGX_CTR_ST_PAIR_T_REFILL], cnt[PGX_CTR_ST_PAIR_T_TOTAL], cnt[PGX_CTR_ST_PAIR_REFILLS],
                     100.0 * (double)cnt[PGX_CTR_ST_PAIR_T_SEED] / (double)cnt[PGX_CTR_ST_PAIR_T_TOTAL], 100.0 * (double)cnt[PGX_CTR_ST_PAIR_T_LINE] / (double)cnt[PGX_CTR_ST_PAIR_T_TOTAL]);
    if (std::getenv("PGX_DEBUG_COUNTERS"))
        std::fprintf(stderr, "[pgx] counters: extensions %llu tag overflows %llu heavy %llu other-image steps %llu lines %llu + %llu seeds %llu + %llu\n", cnt[PGX_CTR_EXT], cnt[PGX_CTR_TAG_OVERFLOW],
                     cnt[PGX_CTR_HEAVY], cnt[PGX_CTR_REDO], cnt[PGX_CTR_PAIRS_LINES], cnt[PGX_CTR_FM_LINES], cnt[PGX_CTR_PAIRS_SEEDS], cnt[PGX_CTR_FM_SEEDS]);
    b->n_ext = n_ext_host;
    b->n_tag_overflow = cnt[PGX_CTR_TAG_OVERFLOW];
    if (b->timed) {
        auto el = [&](int a, int c) { float ms = 0; HIPCHECK(hipEventElapsedTime(&ms, b->ev[a], b->ev[c])); return ms; };
        b->timing.ms_find_mems = chunks.size() > 1 ? ms_fm : el(1, 2);
        b->timing.ms_find_mems_main = chunks.size() > 1 ? ms_main : (n ? el(1, 8) : 0.0f);
        b->timing.ms_compact = chunks.size() > 1 ? ms_cp : el(2, 3);
        if (want_tags) {
            b->timing.ms_tag_locate = el(3, 4); // locate + scans
            b->timing.ms_tag_gather = el(4, 5); // 16-lane small path (gather + sort + unique)
            b->timing.ms_tag_sort = el(5, 6);   // big path + final scan + compaction
        }
        b->timing.ms_total = el(0, 7);
        // what a fresh batch pays before its first find_mems launch (one chunk on the side-stream path: where the passes are)
        if (fresh_work) b->timing.ms_per_upload = b->ms_upload_passes + (fresh_mark ? el(0, 9) : 0.0f);
    }
    if (fresh_work) b->ms_upload_passes = 0; // (reported once)
    break;
    } // (speculative pass, then at most one exact pass)
    b->ran = true;
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_batch_device_result(pgx_batch *b, pgx_device_result *out) {
    PGX_GUARD_BEGIN
    if (!b || !out || !b->ran) throw Error(PGX_ERR_ARG, "pgx_batch_device_result: batch has not been run");
    std::memset(out, 0, sizeof *out);
    out->n_reads = b->n_reads;
    out->n_mems = b->n_mems;
    out->mem_offsets = b->mem_off.as<uint64_t>();
    out->mems = b->mems.as<pgx_mem>();
    if (b->ran_tags) {
        out->n_positions = b->n_positions;
        out->tag_run_counts = b->tw.run_nums.as<uint64_t>();
        out->pos_offsets = b->tw.pos_off.as<uint64_t>();
        out->positions = b->tw.positions.as<uint64_t>();
    }
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_batch_counts(pgx_batch *b, uint64_t *n_mems, uint64_t *n_positions, uint64_t *n_extensions) {
    PGX_GUARD_BEGIN
    if (!b || !b->ran) throw Error(PGX_ERR_ARG, "pgx_batch_counts: batch has not been run");
    if (n_mems) *n_mems = b->n_mems;
    if (n_positions) *n_positions = b->n_positions;
    if (n_extensions) *n_extensions = b->n_ext;
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_batch_spec_stats(pgx_batch *b, uint32_t *speculative_runs, uint32_t *fallbacks) {
    PGX_GUARD_BEGIN
    if (!b) throw Error(PGX_ERR_ARG, "pgx_batch_spec_stats: null batch");
    if (speculative_runs) *speculative_runs = b->spec_runs;
    if (fallbacks) *fallbacks = b->spec_fallbacks;
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_batch_timing(pgx_batch *b, pgx_timing *out) {
    PGX_GUARD_BEGIN
    if (!b || !out || !b->ran) throw Error(PGX_ERR_ARG, "pgx_batch_timing: batch has not been run");
    if (!b->timed) throw Error(PGX_ERR_ARG, "pgx_batch_timing: run without PGX_RUN_TIMING");
    *out = b->timing;
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_batch_result(pgx_batch *b, pgx_result *out) {
    PGX_GUARD_BEGIN
    RoctxRange range("pgx_batch_result");
    if (!b || !out || !b->ran) throw Error(PGX_ERR_ARG, "pgx_batch_result: batch has not been run");
    use_device(b->device);
    const uint64_t n = b->n_reads, m = b->n_mems;
    b->h_mem_off.ensure((n + 1) * 8);
    HIPCHECK(hipMemcpyAsync(b->h_mem_off.p, b->mem_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, b->own));
    b->h_mems.ensure((m ? m : 1) * sizeof(pgx_mem));
    if (m) HIPCHECK(hipMemcpyAsync(b->h_mems.p, b->mems.p, m * sizeof(pgx_mem), hipMemcpyDeviceToHost, b->own));
    std::memset(out, 0, sizeof *out);
    out->n_reads = n;
    out->n_mems = m;
    out->mem_offsets = b->h_mem_off.as<uint64_t>();
    out->mems = b->h_mems.as<pgx_mem>();
    out->n_extensions = b->n_ext;
    if (b->ran_tags) {
        b->h_run_nums.ensure((m ? m : 1) * 8);
        b->h_pos_off.ensure((m + 1) * 8);
        b->h_positions.ensure((b->n_positions ? b->n_positions : 1) * 8);
        if (m) HIPCHECK(hipMemcpyAsync(b->h_run_nums.p, b->tw.run_nums.p, m * 8, hipMemcpyDeviceToHost, b->own));
        HIPCHECK(hipMemcpyAsync(b->h_pos_off.p, b->tw.pos_off.p, (m + 1) * 8, hipMemcpyDeviceToHost, b->own));
        if (b->n_positions) HIPCHECK(hipMemcpyAsync(b->h_positions.p, b->tw.positions.p, b->n_positions * 8, hipMemcpyDeviceToHost, b->own));
        out->tag_run_counts = b->h_run_nums.as<uint64_t>();
        out->pos_offsets = b->h_pos_off.as<uint64_t>();
        out->positions = b->h_positions.as<uint64_t>();
        out->n_positions = b->n_positions;
        out->n_tag_overflow = b->n_tag_overflow;
    }
    HIPCHECK(hipStreamSynchronize(b->own)); // (the run itself completed inside pgx_batch_run, on whatever stream it used)
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_find_mems_batch(pgx_index *h, int device, const uint8_t *reads, const uint64_t *offsets,
                                          uint64_t n_reads, uint64_t min_len, uint64_t min_occ, uint32_t flags,
                                          pgx_batch **batch_out, pgx_result *result_out) {
    if (!batch_out || !result_out) { pgx::set_last_error("pgx_find_mems_batch: null argument"); return PGX_ERR_ARG; }
    *batch_out = nullptr;
    pgx_batch *b = nullptr;
    pgx_status st = pgx_batch_create(h, device, reads, offsets, n_reads, &b);
    if (st == PGX_OK) st = pgx_batch_run(b, min_len, min_occ, flags, nullptr);
    if (st == PGX_OK) st = pgx_batch_result(b, result_out);
    if (st != PGX_OK) { pgx_batch_free(b); return st; }
    *batch_out = b;
    return PGX_OK;
}

// reads sharded over devices: contiguous slices, one host thread + batch + stream per slice, the index image
// replicated per device, no collective; slice i covers reads [first_read[i], first_read[i + 1])
extern "C" pgx_status pgx_find_mems_sharded(pgx_index *h, const int *devices, uint32_t n_slices, const uint8_t *reads, const uint64_t *offsets,
                                            uint64_t n_reads, uint64_t min_len, uint64_t min_occ, uint32_t flags, pgx_batch **batches_out,
                                            pgx_result *results_out, uint64_t *first_read) {
    PGX_GUARD_BEGIN
    if (!h || !devices || !n_slices || !offsets || !batches_out || !results_out || !first_read)
        throw Error(PGX_ERR_ARG, "pgx_find_mems_sharded: null argument");
    for (uint32_t i = 0; i < n_slices; i++) { batches_out[i] = nullptr; first_read[i] = n_reads * i / n_slices; }
    first_read[n_slices] = n_reads;
    for (uint32_t i = 0; i < n_slices; i++) (void)device_image(h, devices[i]); // images first: one upload per distinct device
    std::vector<pgx_status> st(n_slices, PGX_OK);
    std::vector<std::string> err(n_slices);
    std::vector<std::thread> th;
    for (uint32_t i = 0; i < n_slices; i++)
        th.emplace_back([&, i]() {
            const uint64_t a = first_read[i], b = first_read[i + 1];
            st[i] = pgx_find_mems_batch(h, devices[i], reads, offsets + a, b - a, min_len, min_occ, flags, &batches_out[i], &results_out[i]);
            if (st[i] != PGX_OK) err[i] = pgx_last_error(); // (the message is thread-local)
        });
    for (auto &t : th) t.join();
    for (uint32_t i = 0; i < n_slices; i++)
        if (st[i] != PGX_OK) {
            for (uint32_t k = 0; k < n_slices; k++) { pgx_batch_free(batches_out[k]); batches_out[k] = nullptr; }
            throw Error(st[i], "slice " + std::to_string(i) + " (device " + std::to_string(devices[i]) + "): " + err[i]);
        }
    return PGX_OK;
    PGX_GUARD_END
}

// ------------------------------------------------------------------------------------------
// primitives (tests)
extern "C" pgx_status pgx_rank_batch(pgx_index *h, int device, const uint64_t *pos, uint64_t n, int true_codes, uint64_t *out) {
    PGX_GUARD_BEGIN
    if (!h || (n && (!pos || !out))) throw Error(PGX_ERR_ARG, "pgx_rank_batch: null argument");
    if (!h->has_rank) throw Error(PGX_ERR_ARG, "pgx_rank_batch: index opened without an r-index");
    pgx_device_image *d = device_image(h, device);
    if (!n) return PGX_OK;
    DevBuf dp, dout;
    try {
        dp.ensure(n * 8);
        dout.ensure(n * 48);
        HIPCHECK(hipMemcpy(dp.p, pos, n * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemset(dout.p, 0, n * 48));
        const char *probe = std::getenv("PGX_RANK_PROBE"); // scripts/anomaly_probe.py: the round-3 shapes of this kernel (wide dense2 image, true codes)
        if (probe && d->img.dense == 3 && true_codes) {
            const std::string v(probe);
            if (v == "loop_mulhi") hipLaunchKernelGGL((pgx_rank_probe_kernel<true, true>), dim3(grid_for(n, 256)), dim3(256), 0, 0, d->img, dp.as<uint64_t>(), n, dout.as<uint64_t>());
            else if (v == "loop") hipLaunchKernelGGL((pgx_rank_probe_kernel<true, false>), dim3(grid_for(n, 256)), dim3(256), 0, 0, d->img, dp.as<uint64_t>(), n, dout.as<uint64_t>());
            else if (v == "mulhi") hipLaunchKernelGGL((pgx_rank_probe_kernel<false, true>), dim3(grid_for(6 * n, 256)), dim3(256), 0, 0, d->img, dp.as<uint64_t>(), n, dout.as<uint64_t>());
            else throw Error(PGX_ERR_ARG, "PGX_RANK_PROBE: loop_mulhi | loop | mulhi");
        } else
        hipLaunchKernelGGL(pgx_rank_kernel, dim3(grid_for(6 * n, 256)), dim3(256), 0, 0, d->img, dp.as<uint64_t>(), n, true_codes, dout.as<uint64_t>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpy(out, dout.p, n * 48, hipMemcpyDeviceToHost));
    } catch (...) { dp.release(); dout.release(); throw; }
    dp.release(); dout.release();
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_extend_batch(pgx_index *h, int device, const pgx_biint *in, const uint8_t *sym, const uint8_t *forward,
                                       uint64_t n, pgx_biint *out) {
    PGX_GUARD_BEGIN
    if (!h || (n && (!in || !sym || !forward || !out))) throw Error(PGX_ERR_ARG, "pgx_extend_batch: null argument");
    if (!h->has_rank) throw Error(PGX_ERR_ARG, "pgx_extend_batch: index opened without an r-index");
    pgx_device_image *d = device_image(h, device);
    if (!n) return PGX_OK;
    DevBuf din, dsym, dfw, dout;
    try {
        din.ensure(n * sizeof(pgx_biint)); dsym.ensure(n); dfw.ensure(n); dout.ensure(n * sizeof(pgx_biint));
        HIPCHECK(hipMemcpy(din.p, in, n * sizeof(pgx_biint), hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(dsym.p, sym, n, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(dfw.p, forward, n, hipMemcpyHostToDevice));
        if (d->lds_bytes)
            hipLaunchKernelGGL(pgx_extend_kernel<true>, dim3(grid_for(n, 256)), dim3(256), d->lds_bytes, 0, d->img, din.as<pgx_biint>(),
                               dsym.as<uint8_t>(), dfw.as<uint8_t>(), n, dout.as<pgx_biint>());
        else
            hipLaunchKernelGGL(pgx_extend_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, 0, d->img, din.as<pgx_biint>(),
                               dsym.as<uint8_t>(), dfw.as<uint8_t>(), n, dout.as<pgx_biint>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpy(out, dout.p, n * sizeof(pgx_biint), hipMemcpyDeviceToHost));
    } catch (...) { din.release(); dsym.release(); dfw.release(); dout.release(); throw; }
    din.release(); dsym.release(); dfw.release(); dout.release();
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_count_batch(pgx_index *h, int device, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads,
                                      pgx_range *out) {
    PGX_GUARD_BEGIN
    if (!h || !offsets || (n_reads && (!out || (!reads && offsets[n_reads] != offsets[0]))))
        throw Error(PGX_ERR_ARG, "pgx_count_batch: null argument");
    if (!h->has_rank) throw Error(PGX_ERR_ARG, "pgx_count_batch: index opened without an r-index");
    const bool lit = literal_count(h);
    pgx_device_image *d = lit ? literal_image(h, device) : device_image(h, device);
    if (!n_reads) return PGX_OK;
    DevBuf dr, doff, dout;
    try {
        const uint64_t lo = offsets[0], bytes = offsets[n_reads] - lo;
        std::vector<uint64_t> reb(n_reads + 1);
        for (uint64_t i = 0; i <= n_reads; i++) {
            if (i && offsets[i] < offsets[i - 1]) throw Error(PGX_ERR_ARG, "pgx_count_batch: offsets must be non-decreasing");
            reb[i] = offsets[i] - lo;
        }
        dr.ensure(bytes + 16); doff.ensure((n_reads + 1) * 8); dout.ensure(n_reads * sizeof(pgx_range));
        if (bytes) HIPCHECK(hipMemcpy(dr.p, reads + lo, bytes, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(doff.p, reb.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice));
        if (lit)
            hipLaunchKernelGGL(pgx_lit_count_kernel, dim3(grid_for(n_reads, 256)), dim3(256), 0, 0, d->lit, (const uint8_t *)dr.as<uint8_t>(),
                               (const uint64_t *)doff.as<uint64_t>(), (const pgx_range *)nullptr, (const uint8_t *)nullptr, n_reads, dout.as<pgx_range>());
        else if (d->lds_bytes)
            hipLaunchKernelGGL(pgx_count_kernel<true>, dim3(grid_for(n_reads, 256)), dim3(256), d->lds_bytes, 0, d->img, dr.as<uint8_t>(),
                               doff.as<uint64_t>(), n_reads, dout.as<pgx_range>());
        else
            hipLaunchKernelGGL(pgx_count_kernel<false>, dim3(grid_for(n_reads, 256)), dim3(256), 0, 0, d->img, dr.as<uint8_t>(),
                               doff.as<uint64_t>(), n_reads, dout.as<pgx_range>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpy(out, dout.p, n_reads * sizeof(pgx_range), hipMemcpyDeviceToHost));
    } catch (...) { dr.release(); doff.release(); dout.release(); throw; }
    dr.release(); doff.release(); dout.release();
    return PGX_OK;
    PGX_GUARD_END
}

// COMPAT count_encoded / LF_encoded on an encoded index without N: the reference mis-parses every block (quirk 3); the literal
// image reproduces what it computes
static bool literal_count(const pgx_index *h) { return (h->mode & PGX_MODE_MASK) == PGX_MODE_COMPAT && h->ri.encoded && !h->ri.hasN; }
static pgx_device_image *literal_image(pgx_index *h, int device) {
    pgx_device_image *d = device_image(h, device);
    std::lock_guard<std::mutex> lock(g_image_mutex);
    if (d->has_lit) return d;
    build_literal_image(h->ri, h->lit);
    LitHostImage &m = h->lit;
    upload(d->lit_bstart, m.bstart.data(), m.bstart.size() * 8);
    upload(d->lit_cum, m.cum.data(), m.cum.size() * 8);
    upload(d->lit_runs, m.runs.data(), m.runs.size() * 8);
    upload(d->lit_roff, m.roff.data(), m.roff.size() * 4);
    d->lit_tabs.ensure(512 * 4);
    HIPCHECK(hipMemcpy(d->lit_tabs.p, m.code_of, 1024, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d->lit_tabs.as<uint32_t>() + 256, m.cslot_of, 1024, hipMemcpyHostToDevice));
    PgxLitImage &g = d->lit;
    g.bstart = d->lit_bstart.as<uint64_t>(); g.cum = d->lit_cum.as<uint64_t>(); g.runs = d->lit_runs.as<uint64_t>();
    g.roff = d->lit_roff.as<uint32_t>(); g.code_of = d->lit_tabs.as<uint32_t>(); g.cslot_of = d->lit_tabs.as<uint32_t>() + 256;
    for (int i = 0; i < 8; i++) g.C[i] = m.C[i];
    g.n = h->ri.sequence_size; g.n_blocks = m.bstart.size();
    d->has_lit = true;
    return d;
}

extern "C" pgx_status pgx_find_mems_function_batch(pgx_index *h, int device, const uint8_t *reads, const uint64_t *offsets, uint64_t n_reads,
                                                    const uint64_t *read_of, const uint64_t *x, uint64_t n, uint64_t min_len, uint64_t min_occ,
                                                    uint64_t *next_x, pgx_mem *mem, uint8_t *has_mem, uint64_t *n_ext) {
    PGX_GUARD_BEGIN
    if (!h || !offsets || (n && (!read_of || !x || !next_x || !mem || !has_mem))) throw Error(PGX_ERR_ARG, "pgx_find_mems_function_batch: null argument");
    if (!h->has_rank) throw Error(PGX_ERR_ARG, "pgx_find_mems_function_batch: index opened without an r-index");
    pgx_device_image *d = device_image(h, device);
    if (!n) return PGX_OK;
    for (uint64_t i = 0; i < n_reads; i++) {
        if (offsets[i + 1] < offsets[i]) throw Error(PGX_ERR_ARG, "pgx_find_mems_function_batch: offsets must be non-decreasing");
        if (offsets[i + 1] - offsets[i] >= (1ull << 31)) throw Error(PGX_ERR_UNSUPPORTED, "read longer than 2^31 bytes");
    }
    for (uint64_t i = 0; i < n; i++)
        if (read_of[i] >= n_reads) throw Error(PGX_ERR_ARG, "pgx_find_mems_function_batch: read index out of range");
    DevBuf dr, doff, dro, dx, dout;
    DevBuf *all[] = {&dr, &doff, &dro, &dx, &dout};
    try {
        const uint64_t lo = offsets[0], bytes = offsets[n_reads] - lo;
        if (bytes && !reads) throw Error(PGX_ERR_ARG, "pgx_find_mems_function_batch: null reads");
        std::vector<uint64_t> reb(n_reads + 1);
        for (uint64_t i = 0; i <= n_reads; i++) reb[i] = offsets[i] - lo;
        dr.ensure(bytes + 32); doff.ensure((n_reads + 1) * 8); dro.ensure(n * 8); dx.ensure(n * 8); dout.ensure(n * sizeof(PgxHeavyResult));
        HIPCHECK(hipMemset((uint8_t *)dr.p + bytes, 0, 32));
        if (bytes) HIPCHECK(hipMemcpy(dr.p, reads + lo, bytes, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(doff.p, reb.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(dro.p, read_of, n * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(dx.p, x, n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(pgx_fmf_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, d->img, dr.as<uint8_t>(), doff.as<uint64_t>(), dro.as<uint64_t>(),
                           dx.as<uint64_t>(), n, min_len, min_occ, dout.as<PgxHeavyResult>());
        HIPCHECK(hipGetLastError());
        std::vector<PgxHeavyResult> res(n);
        HIPCHECK(hipMemcpy(res.data(), dout.p, n * sizeof(PgxHeavyResult), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; i++) {
            next_x[i] = res[i].next_x; mem[i] = res[i].mem; has_mem[i] = (uint8_t)res[i].has_mem;
            if (n_ext) n_ext[i] = res[i].n_ext;
        }
    } catch (...) { for (DevBuf *b : all) b->release(); throw; }
    for (DevBuf *b : all) b->release();
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_lf_batch(pgx_index *h, int device, const pgx_range *in, const uint8_t *sym, uint64_t n, pgx_range *out) {
    PGX_GUARD_BEGIN
    if (!h || (n && (!in || !sym || !out))) throw Error(PGX_ERR_ARG, "pgx_lf_batch: null argument");
    if (!h->has_rank) throw Error(PGX_ERR_ARG, "pgx_lf_batch: index opened without an r-index");
    const bool lit = literal_count(h);
    pgx_device_image *d = lit ? literal_image(h, device) : device_image(h, device);
    if (!n) return PGX_OK;
    DevBuf din, dsym, dout;
    try {
        din.ensure(n * sizeof(pgx_range)); dsym.ensure(n); dout.ensure(n * sizeof(pgx_range));
        HIPCHECK(hipMemcpy(din.p, in, n * sizeof(pgx_range), hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(dsym.p, sym, n, hipMemcpyHostToDevice));
        if (lit)
            hipLaunchKernelGGL(pgx_lit_count_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, d->lit, (const uint8_t *)nullptr, (const uint64_t *)nullptr,
                               (const pgx_range *)din.as<pgx_range>(), (const uint8_t *)dsym.as<uint8_t>(), n, dout.as<pgx_range>());
        else
            hipLaunchKernelGGL(pgx_lf_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, d->img, din.as<pgx_range>(), dsym.as<uint8_t>(), n, dout.as<pgx_range>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpy(out, dout.p, n * sizeof(pgx_range), hipMemcpyDeviceToHost));
    } catch (...) { din.release(); dsym.release(); dout.release(); throw; }
    din.release(); dsym.release(); dout.release();
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_tag_query_batch(pgx_index *h, int device, const uint64_t *start, const uint64_t *end, uint64_t n,
                                          uint64_t *run_nums, uint64_t *pos_offsets, uint64_t *positions, uint64_t positions_cap,
                                          uint64_t *n_overflow) {
    PGX_GUARD_BEGIN
    if (!h || !pos_offsets || (n && (!start || !end || !run_nums))) throw Error(PGX_ERR_ARG, "pgx_tag_query_batch: null argument");
    if (!h->has_tags) throw Error(PGX_ERR_ARG, "pgx_tag_query_batch: index opened without a tag array");
    pgx_device_image *d = device_image(h, device);
    pos_offsets[0] = 0;
    if (n_overflow) *n_overflow = 0;
    if (!n) return PGX_OK;
    DevBuf ds, de, dctr;
    TagWork w;
    try {
        hipStream_t s = nullptr;
        ds.ensure(n * 8); de.ensure(n * 8); dctr.ensure(256);
        HIPCHECK(hipMemset(dctr.p, 0, 256));
        HIPCHECK(hipMemcpy(ds.p, start, n * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(de.p, end, n * 8, hipMemcpyHostToDevice));
        unsigned long long *ctr = dctr.as<unsigned long long>();
        tag_pipeline(d->img, nullptr, ds.as<uint64_t>(), de.as<uint64_t>(), n, w, ctr, ctr + 1, s, [](int) {});
        HIPCHECK(hipMemcpy(run_nums, w.run_nums.p, n * 8, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(pos_offsets, w.pos_off.p, (n + 1) * 8, hipMemcpyDeviceToHost));
        if (positions) {
            if (positions_cap < w.n_positions) throw Error(PGX_ERR_ARG, "pgx_tag_query_batch: positions_cap too small");
            if (w.n_positions) HIPCHECK(hipMemcpy(positions, w.positions.p, w.n_positions * 8, hipMemcpyDeviceToHost));
        }
        if (n_overflow) {
            unsigned long long c = 0;
            HIPCHECK(hipMemcpy(&c, dctr.p, 8, hipMemcpyDeviceToHost));
            *n_overflow = c;
        }
    } catch (...) {
        ds.release(); de.release(); dctr.release(); w.release();
        throw;
    }
    ds.release(); de.release(); dctr.release(); w.release();
    return PGX_OK;
    PGX_GUARD_END
}

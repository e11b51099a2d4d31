// pgx_exchange.hip -- chromosome-sharded mode (BASELINE configs[4]): the one place the find_mems path has a real
// exchange step.  Every rank holds the indexes of some chromosomes ("shards"), every read is searched in every shard, and the
// per-read result is the concatenation, in shard order, of the per-shard MEM lists.  The per-read offsets (u32) and the 32-byte
// MEM records travel between ranks over RCCL (xGMI), straight from and into device memory:
//   ncclAllGather   local CSR offsets: max_local x (n_reads + 1) u32 per rank (slots of ranks with fewer shards stay zero)
//   ncclBroadcast   one per rank inside one group, each exactly as long as that rank's record list (no padding to the
//                   largest rank)
// then one kernel sums the per-read totals, a device scan places the reads, and one kernel interleaves the records of every
// read in shard order.  RCCL is loaded on first use (dlopen): single-GPU users of libpgx do not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "pgx_device.h"
#include "pgx_host.hpp"
#include "pgx_runtime.hpp"

using namespace pgx;

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, []() {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); // (a process that already holds an RCCL -- torch's -- gets that one back)
            if (r.lib) break;
        }
        if (!r.lib) { err = std::string("RCCL is not available: ") + dlerror(); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p && err.empty()) err = std::string("RCCL symbol missing: ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!err.empty()) throw Error(PGX_ERR_UNSUPPORTED, err);
    return r;
}
#define NCCLCHECK(expr)                                                                                        \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) throw Error(PGX_ERR_HIP, std::string(#expr) + " failed: " + rccl().GetErrorString(r_)); \
    } while (0)
} // namespace

struct pgx_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    DevBuf local_offs, all_offs, send, all_recs, totals, out_offs, out_mems, out_shard, meta, scan_tmp;
    uint64_t n_reads = 0, n_mems = 0;
};

// local CSR offsets of one shard as u32 (a batch holds far fewer than 2^32 MEMs: its reads are chunked by the caller)
__global__ void __launch_bounds__(256) pgx_xch_offsets_kernel(const uint64_t *__restrict__ mem_off, uint64_t n1, uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) out[i] = (uint32_t)mem_off[i];
}
// per-read total over all shards; slot (r, k) of the gathered offsets = shard_slot[c]
__global__ void __launch_bounds__(256)
pgx_xch_totals_kernel(const uint32_t *__restrict__ all_offs, const uint32_t *__restrict__ shard_slot, uint32_t n_shards, uint64_t n_reads,
                      uint64_t *__restrict__ totals) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    uint64_t t = 0;
    for (uint32_t c = 0; c < n_shards; c++) {
        const uint32_t *o = all_offs + (size_t)shard_slot[c] * (n_reads + 1);
        t += o[i + 1] - o[i];
    }
    totals[i] = t;
}
// the records of read i: shard 0's, then shard 1's, ... ; src_base[c] = first record of shard c in the gathered record array
__global__ void __launch_bounds__(256)
pgx_xch_interleave_kernel(const uint32_t *__restrict__ all_offs, const uint32_t *__restrict__ shard_slot, const uint64_t *__restrict__ src_base,
                          uint32_t n_shards, uint64_t n_reads, const pgx_mem *__restrict__ recs, const uint64_t *__restrict__ out_offs,
                          pgx_mem *__restrict__ out, uint32_t *__restrict__ shard_of) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    uint64_t dst = out_offs[i];
    for (uint32_t c = 0; c < n_shards; c++) {
        const uint32_t *o = all_offs + (size_t)shard_slot[c] * (n_reads + 1);
        const uint32_t a = o[i], b = o[i + 1];
        const pgx_mem *src = recs + src_base[c] + a;
        for (uint32_t t = 0; t < b - a; t++) { out[dst] = src[t]; shard_of[dst] = c; dst++; }
    }
}

extern "C" pgx_status pgx_comm_unique_id(uint8_t id[PGX_COMM_ID_BYTES]) {
    PGX_GUARD_BEGIN
    if (!id) throw Error(PGX_ERR_ARG, "pgx_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == PGX_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCLCHECK(rccl().GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" void pgx_comm_free(pgx_comm *c) {
    if (!c) return;
    if (hipSetDevice(c->device) == hipSuccess) {
        DevBuf *all[] = {&c->local_offs, &c->all_offs, &c->send, &c->all_recs, &c->totals, &c->out_offs, &c->out_mems, &c->out_shard, &c->meta, &c->scan_tmp};
        for (DevBuf *d : all) d->release();
        if (c->comm) (void)rccl().CommDestroy(c->comm);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

extern "C" pgx_status pgx_comm_init(const uint8_t id[PGX_COMM_ID_BYTES], int rank, int world, int device, pgx_comm **out) {
    PGX_GUARD_BEGIN
    if (!id || !out || world < 1 || rank < 0 || rank >= world) throw Error(PGX_ERR_ARG, "pgx_comm_init: bad argument");
    *out = nullptr;
    pgx_use_device(device);
    std::unique_ptr<pgx_comm, void (*)(pgx_comm *)> c(new pgx_comm(), pgx_comm_free);
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    NCCLCHECK(rccl().CommInitRank(&c->comm, world, u, rank));
    HIPCHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c.release();
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" uint64_t pgx_exchange_owner_digest(const uint32_t *owner_of_shard, uint32_t n_shards) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; owner_of_shard && i < n_shards; i++)
        for (int b = 0; b < 4; b++) { h ^= (owner_of_shard[i] >> (8 * b)) & 0xFFu; h *= 0x100000001b3ull; }
    return h;
}

// shards per rank and the largest such count (rows per rank in the gathered offsets)
static uint32_t shards_per_rank(uint32_t world, const uint32_t *owner_of_shard, uint32_t n_shards, std::vector<uint32_t> &per_rank, std::vector<uint32_t> &kth) {
    per_rank.assign(world, 0);
    kth.assign(n_shards, 0);
    for (uint32_t sh = 0; sh < n_shards; sh++) {
        if (owner_of_shard[sh] >= world) throw Error(PGX_ERR_ARG, "pgx_exchange_mems: owner rank out of range");
        kth[sh] = per_rank[owner_of_shard[sh]]++;
    }
    uint32_t max_local = 1;
    for (uint32_t v : per_rank) max_local = std::max(max_local, v);
    return max_local;
}

extern "C" pgx_status pgx_exchange_plan(uint32_t world, const uint32_t *owner_of_shard, uint32_t n_shards, const uint64_t *gathered,
                                        uint32_t *max_local_out, uint32_t *slot, uint64_t *rec_base, uint64_t *src_base, uint64_t *n_reads_out) {
    PGX_GUARD_BEGIN
    if (!world || !owner_of_shard || !n_shards || !gathered || !max_local_out || !slot || !rec_base || !src_base || !n_reads_out)
        throw Error(PGX_ERR_ARG, "pgx_exchange_plan: null argument");
    std::vector<uint32_t> per_rank, kth;
    const uint32_t max_local = shards_per_rank(world, owner_of_shard, n_shards, per_rank, kth);
    const size_t meta_n = PGX_XCH_META_HEAD + max_local;
    const uint64_t digest = pgx_exchange_owner_digest(owner_of_shard, n_shards);
    // every rank sees the same rows, so every rank fails (or not) here together
    for (uint32_t r = 0; r < world; r++) {
        const uint64_t *m = gathered + (size_t)r * meta_n;
        if (m[0]) throw Error(PGX_ERR_ARG, "pgx_exchange_mems: rank " + std::to_string(r) + " reported status " + std::to_string(m[0]) + " (its own error message says why)");
        if (m[2] != n_shards || m[3] != digest || m[4] != max_local) throw Error(PGX_ERR_ARG, "pgx_exchange_mems: rank " + std::to_string(r) + " passed a different owner_of_shard table");
    }
    bool have = false;
    uint64_t n_reads = 0;
    for (uint32_t r = 0; r < world; r++) { // ranks without shards learn n_reads from the ones that have some
        if (!per_rank[r]) continue;
        const uint64_t nr = gathered[(size_t)r * meta_n + 1];
        if (have && nr != n_reads) throw Error(PGX_ERR_ARG, "pgx_exchange_mems: ranks disagree on the number of reads");
        n_reads = nr; have = true;
    }
    if ((uint64_t)world * max_local * (n_reads + 1) * 4 > PGX_XCH_MAX_OFFSET_BYTES)
        throw Error(PGX_ERR_ARG, "pgx_exchange_mems: " + std::to_string(n_reads) + " reads x " + std::to_string(world * max_local) +
                                     " offset rows exceed PGX_XCH_MAX_OFFSET_BYTES: exchange the reads in chunks");
    for (uint32_t sh = 0; sh < n_shards; sh++) slot[sh] = owner_of_shard[sh] * max_local + kth[sh];
    rec_base[0] = 0;
    for (uint32_t r = 0; r < world; r++) {
        uint64_t m = 0;
        for (uint32_t k = 0; k < per_rank[r]; k++) m += gathered[(size_t)r * meta_n + PGX_XCH_META_HEAD + k];
        rec_base[r + 1] = rec_base[r] + m;
    }
    for (uint32_t sh = 0; sh < n_shards; sh++) {
        const uint32_t r = owner_of_shard[sh];
        uint64_t b = rec_base[r];
        for (uint32_t k = 0; k < kth[sh]; k++) b += gathered[(size_t)r * meta_n + PGX_XCH_META_HEAD + k];
        src_base[sh] = b;
    }
    *max_local_out = max_local;
    *n_reads_out = n_reads;
    return PGX_OK;
    PGX_GUARD_END
}

namespace {
struct GroupGuard { // a throw between ncclGroupStart and ncclGroupEnd must not leave the group open
    bool open = false;
    void start() { NCCLCHECK(rccl().GroupStart()); open = true; }
    void end() { open = false; NCCLCHECK(rccl().GroupEnd()); }
    ~GroupGuard() { if (open) (void)rccl().GroupEnd(); }
};
} // namespace

extern "C" pgx_status pgx_exchange_mems(pgx_comm *c, pgx_batch *const *batches, const uint32_t *shard_ids, uint32_t n_local,
                                        const uint32_t *owner_of_shard, uint32_t n_shards, pgx_exchange_result *out) {
    PGX_GUARD_BEGIN
    // Arguments every rank must agree on BEFORE any collective (world, n_shards > 0): a violation here cannot be reported to the peers.
    if (!c || !out || !owner_of_shard || !n_shards) throw Error(PGX_ERR_ARG, "pgx_exchange_mems: null argument");
    pgx_use_device(c->device);
    hipStream_t s = c->stream;
    std::vector<uint32_t> per_rank, kth;
    const uint32_t max_local = shards_per_rank((uint32_t)c->world, owner_of_shard, n_shards, per_rank, kth);
    // Rank-local validation: a failure becomes the status word of this rank's metadata row -- the rank still enters the metadata
    // all-gather, and every rank then fails together in pgx_exchange_plan instead of this one leaving its peers blocked in a collective.
    uint64_t status = 0;
    std::string why;
    auto fail = [&](uint64_t code, const std::string &msg) { if (!status) { status = code; why = msg; } };
    if (n_local && (!batches || !shard_ids)) { fail(PGX_ERR_ARG, "pgx_exchange_mems: null argument"); n_local = 0; }
    if (n_local != per_rank[c->rank])
        fail(PGX_ERR_ARG, "pgx_exchange_mems: this rank passes " + std::to_string(n_local) + " batches but owns " + std::to_string(per_rank[c->rank]) + " shards");
    // local batches in ascending shard order
    std::vector<uint32_t> order(n_local);
    for (uint32_t k = 0; k < n_local; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return shard_ids[a] < shard_ids[b]; });
    uint64_t n_reads = 0;
    std::vector<pgx_device_result> dr(n_local);
    uint64_t m_local = 0;
    for (uint32_t k = 0; k < n_local && !status; k++) {
        const uint32_t b = order[k];
        if (shard_ids[b] >= n_shards || owner_of_shard[shard_ids[b]] != (uint32_t)c->rank || (k && shard_ids[order[k - 1]] == shard_ids[b])) {
            fail(PGX_ERR_ARG, "pgx_exchange_mems: shard ids of this rank do not match owner_of_shard"); break; }
        if (pgx_batch_device_result(batches[b], &dr[k]) != PGX_OK) { fail(PGX_ERR_ARG, std::string("pgx_exchange_mems: ") + pgx_last_error()); break; }
        if (k == 0) n_reads = dr[k].n_reads;
        else if (dr[k].n_reads != n_reads) { fail(PGX_ERR_ARG, "pgx_exchange_mems: every shard must have searched the same reads"); break; }
        if (dr[k].n_mems >> 32) { fail(PGX_ERR_UNSUPPORTED, "pgx_exchange_mems: more than 2^32 MEMs in one shard batch (use smaller read batches)"); break; }
        m_local += dr[k].n_mems;
    }
    // ---- 1. metadata row of this rank: {status, n_reads, n_shards, digest of the owner table, max_local, MEMs of every local shard}, gathered as u64
    const size_t meta_n = PGX_XCH_META_HEAD + max_local;
    std::vector<uint64_t> h_meta(meta_n, 0), h_all_meta((size_t)c->world * meta_n, 0);
    h_meta[0] = status; h_meta[1] = n_reads; h_meta[2] = n_shards; h_meta[3] = pgx_exchange_owner_digest(owner_of_shard, n_shards); h_meta[4] = max_local;
    for (uint32_t k = 0; k < n_local && !status; k++) h_meta[PGX_XCH_META_HEAD + k] = dr[k].n_mems;
    c->meta.ensure((size_t)(c->world + 1) * meta_n * 8);
    uint64_t *d_meta = c->meta.as<uint64_t>(), *d_all_meta = d_meta + meta_n;
    HIPCHECK(hipMemcpyAsync(d_meta, h_meta.data(), meta_n * 8, hipMemcpyHostToDevice, s));
    NCCLCHECK(rccl().AllGather(d_meta, d_all_meta, meta_n, ncclUint64, c->comm, s));
    HIPCHECK(hipMemcpyAsync(h_all_meta.data(), d_all_meta, (size_t)c->world * meta_n * 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (status) throw Error((pgx_status)status, why); // (the peers fail in the plan below, on this rank's status word)
    // ---- the plan: identical on every rank (same gathered rows), so is its verdict
    std::vector<uint32_t> slot(n_shards);
    std::vector<uint64_t> rec_base(c->world + 1, 0), src_base(n_shards, 0);
    uint32_t ml = 0;
    if (pgx_exchange_plan((uint32_t)c->world, owner_of_shard, n_shards, h_all_meta.data(), &ml, slot.data(), rec_base.data(), src_base.data(), &n_reads) != PGX_OK)
        throw Error(PGX_ERR_ARG, pgx_last_error());
    // ---- 2. local offsets (u32) and records (concatenated in shard order)
    const size_t row = n_reads + 1;
    c->local_offs.ensure((size_t)max_local * row * 4);
    c->all_offs.ensure((size_t)c->world * max_local * row * 4);
    HIPCHECK(hipMemsetAsync(c->local_offs.p, 0, (size_t)max_local * row * 4, s));
    c->send.ensure((m_local ? m_local : 1) * sizeof(pgx_mem));
    uint64_t at = 0;
    for (uint32_t k = 0; k < n_local; k++) {
        hipLaunchKernelGGL(pgx_xch_offsets_kernel, dim3((unsigned)((row + 255) / 256)), dim3(256), 0, s, dr[k].mem_offsets, (uint64_t)row,
                           c->local_offs.as<uint32_t>() + (size_t)k * row);
        if (dr[k].n_mems) HIPCHECK(hipMemcpyAsync(c->send.as<pgx_mem>() + at, dr[k].mems, dr[k].n_mems * sizeof(pgx_mem), hipMemcpyDeviceToDevice, s));
        at += dr[k].n_mems;
    }
    HIPCHECK(hipGetLastError());
    NCCLCHECK(rccl().AllGather(c->local_offs.p, c->all_offs.p, (size_t)max_local * row, ncclUint32, c->comm, s));
    // ---- 3. records: one broadcast per rank, each as long as that rank's list
    const uint64_t total = rec_base[c->world];
    c->all_recs.ensure((total ? total : 1) * sizeof(pgx_mem));
    {
        GroupGuard g;
        g.start();
        for (int r = 0; r < c->world; r++) {
            const uint64_t m = rec_base[r + 1] - rec_base[r];
            if (!m) continue;
            NCCLCHECK(rccl().Broadcast(c->send.p, c->all_recs.as<pgx_mem>() + rec_base[r], m * 4, ncclUint64, r, c->comm, s));
        }
        g.end();
    }
    // ---- 4. per-read totals -> scan -> interleave
    c->totals.ensure((n_reads ? n_reads : 1) * 8);
    c->out_offs.ensure((n_reads + 1) * 8);
    c->out_mems.ensure((total ? total : 1) * sizeof(pgx_mem));
    c->out_shard.ensure((total ? total : 1) * 4);
    // small tables for the kernels, after the gathered metadata in c->meta
    const size_t tab_bytes = (size_t)n_shards * 4 + (size_t)n_shards * 8 + 16;
    DevBuf &tab = c->scan_tmp; // (scan scratch lives behind the tables)
    const size_t scan_need = ((n_reads + PGX_SCAN_BLOCK_ITEMS - 1) / PGX_SCAN_BLOCK_ITEMS + 2) * 8;
    tab.ensure(tab_bytes + scan_need + 64);
    uint64_t *d_src_base = tab.as<uint64_t>();
    uint32_t *d_slot = reinterpret_cast<uint32_t *>(d_src_base + n_shards);
    uint64_t *d_scan = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(tab.p) + ((tab_bytes + 63) & ~(size_t)63));
    HIPCHECK(hipMemcpyAsync(d_src_base, src_base.data(), (size_t)n_shards * 8, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(d_slot, slot.data(), (size_t)n_shards * 4, hipMemcpyHostToDevice, s));
    if (n_reads) {
        hipLaunchKernelGGL(pgx_xch_totals_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, s, c->all_offs.as<uint32_t>(), (const uint32_t *)d_slot,
                           n_shards, n_reads, c->totals.as<uint64_t>());
        HIPCHECK(hipGetLastError());
    }
    pgx_scan_u64(c->totals.as<uint64_t>(), n_reads, c->out_offs.as<uint64_t>(), d_scan, s);
    if (n_reads) {
        hipLaunchKernelGGL(pgx_xch_interleave_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, s, c->all_offs.as<uint32_t>(), (const uint32_t *)d_slot,
                           (const uint64_t *)d_src_base, n_shards, n_reads, c->all_recs.as<pgx_mem>(), c->out_offs.as<uint64_t>(),
                           c->out_mems.as<pgx_mem>(), c->out_shard.as<uint32_t>());
        HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipStreamSynchronize(s)); // (host vectors above were sources of asynchronous copies)
    c->n_reads = n_reads;
    c->n_mems = total;
    out->n_reads = n_reads;
    out->n_mems = total;
    out->mem_offsets = c->out_offs.as<uint64_t>();
    out->mems = c->out_mems.as<pgx_mem>();
    out->shard_of_mem = c->out_shard.as<uint32_t>();
    return PGX_OK;
    PGX_GUARD_END
}

extern "C" pgx_status pgx_exchange_download(pgx_comm *c, uint64_t *mem_offsets, pgx_mem *mems, uint32_t *shard_of_mem) {
    PGX_GUARD_BEGIN
    if (!c) throw Error(PGX_ERR_ARG, "pgx_exchange_download: null argument");
    pgx_use_device(c->device);
    if (mem_offsets) HIPCHECK(hipMemcpy(mem_offsets, c->out_offs.p, (c->n_reads + 1) * 8, hipMemcpyDeviceToHost));
    if (mems && c->n_mems) HIPCHECK(hipMemcpy(mems, c->out_mems.p, c->n_mems * sizeof(pgx_mem), hipMemcpyDeviceToHost));
    if (shard_of_mem && c->n_mems) HIPCHECK(hipMemcpy(shard_of_mem, c->out_shard.p, c->n_mems * 4, hipMemcpyDeviceToHost));
    return PGX_OK;
    PGX_GUARD_END
}

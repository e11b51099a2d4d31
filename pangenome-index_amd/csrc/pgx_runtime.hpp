// pgx_runtime.hpp -- what the HIP translation units of libpgx share: error macros, grow-only device / pinned buffers,
// device selection and the device-wide exclusive scan.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>

#include "pgx_host.hpp"

using pgx::Error;

#define HIPCHECK(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            throw Error(PGX_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(e_));    \
    } while (0)

#define PGX_GUARD_BEGIN try {
#define PGX_GUARD_END                                                                               \
    }                                                                                               \
    catch (const pgx::Error &e) { pgx::set_last_error(e.what()); return e.code; }                   \
    catch (const std::bad_alloc &) { pgx::set_last_error("out of host memory"); return PGX_ERR_NOMEM; } \
    catch (const std::exception &e) { pgx::set_last_error(e.what()); return PGX_ERR_HIP; }

struct DevBuf { // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
    uint32_t scan_epoch = 0; // pgx_scan_onepass_kernel's tile words in this buffer: epoch of the last scan; 0 = the memory has not been cleared yet
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        scan_epoch = 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            throw Error(PGX_ERR_NOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
        }
        cap = want;
    }
    // grow while keeping the first `keep` bytes (device-to-device copy)
    void ensure_keep(size_t bytes, size_t keep) {
        if (bytes <= cap) return;
        void *old = p;
        size_t want = bytes + bytes / 2 + 256;
        void *np = nullptr;
        hipError_t e = hipMalloc(&np, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            throw Error(PGX_ERR_NOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
        }
        if (old && keep) {
            e = hipMemcpy(np, old, keep, hipMemcpyDeviceToDevice);
            if (e != hipSuccess) { (void)hipFree(np); throw Error(PGX_ERR_HIP, std::string("hipMemcpy failed: ") + hipGetErrorString(e)); }
        }
        if (old) (void)hipFree(old);
        p = np; cap = want;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct HostBuf { // grow-only pinned host buffer (fast D2H; returned to the caller as result arrays)
    void *p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            throw Error(PGX_ERR_NOMEM, "hipHostMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
        }
        cap = want;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};


void pgx_use_device(int device); // hipSetDevice after checking that the ordinal exists (PGX_ERR_NO_DEVICE / PGX_ERR_ARG)
// exclusive scan of n u64 values on stream s: out has n + 1 entries (out[n] = total); tmp holds (n / 2048 + 2) u64 of scratch
void pgx_scan_u64(const uint64_t *in, uint64_t n, uint64_t *out, uint64_t *tmp, hipStream_t s);

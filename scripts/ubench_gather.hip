// ubench_gather.hip -- ceiling for the access pattern of pgx_find_mems_kernel: every lane reads one
// random, aligned B-byte record (B = 64 or 128) from a table, as 16-byte loads, with a dependent
// address chain (next index = hash of the data just read), many waves in flight.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather.hip -o gpurun_out/ubench_gather && gpurun_out/ubench_gather
//
// Prints records/s and GB/s per (table size, record size, waves per SIMD).  Used in DESIGN.md section 5
// to price the kernel against what the memory system delivers for this pattern (not a product file).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int REC16> // record size in 16-byte units (4 = 64 B, 8 = 128 B)
__global__ void __launch_bounds__(256) gather(const uint4 *__restrict__ tab, uint64_t n_rec, int iters, uint64_t *__restrict__ out) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27; // xorshift64*
        const uint64_t r = (x * 0x2545F4914F6CDD1Dull) % n_rec;
        const uint4 *p = tab + r * REC16;
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < REC16; k++) { const uint4 v = p[k]; s += v.x ^ v.y ^ v.z ^ v.w; }
        acc += s;
        x += s; // dependent chain like a rank probe feeding the next position
    }
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    // default table sizes, or the ones named on the command line (MB): the whole-genome-scale PAIRS image is 5.8 GB, its seed table 16 GiB
    std::vector<size_t> sizes_mb = {4, 36, 300, 2048};
    if (argc > 1) { sizes_mb.clear(); for (int i = 1; i < argc; i++) sizes_mb.push_back((size_t)atoll(argv[i])); }
    const int iters = 400;
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20;
        uint4 *tab = nullptr;
        CHECK(hipMalloc(&tab, bytes));
        CHECK(hipMemset(tab, 1, bytes));
        for (int rec16 : {4, 8}) {
            for (int wps : {2, 4, 8}) { // waves per SIMD = blocks per CU (256-thread blocks = 1 wave per SIMD)
                const int grid = cus * wps;
                uint64_t *out = nullptr;
                CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
                hipEvent_t a, b;
                CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
                const uint64_t n_rec = bytes / (16 * rec16);
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(a, 0));
                    if (rec16 == 4) hipLaunchKernelGGL(gather<4>, dim3(grid), dim3(256), 0, 0, tab, n_rec, iters, out);
                    else hipLaunchKernelGGL(gather<8>, dim3(grid), dim3(256), 0, 0, tab, n_rec, iters, out);
                    CHECK(hipEventRecord(b, 0));
                    CHECK(hipEventSynchronize(b));
                }
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, a, b));
                const double recs = (double)grid * 256 * iters;
                printf("table %5zu MB  record %3d B  waves/SIMD %d : %7.2f G records/s  %7.1f GB/s  (%.3f ms)\n", mb, rec16 * 16, wps,
                       recs / ms / 1e6, recs * rec16 * 16 / ms / 1e6, ms);
                CHECK(hipFree(out));
            }
        }
        CHECK(hipFree(tab));
    }
    return 0;
}

// ubench_gather_pairs.hip -- what a rank probe of TWO lines costs, by where the second line is: dependent random gathers from a 1.3 GB / 2.6 GB table,
// each step reading five 16-byte pieces per line of (a) one 128-byte line, (b) two ADJACENT lines (one aligned 256-byte block), (c) two independent
// random lines.  Decides whether a 256-byte PAIRS block (352 positions: intervals of ~100 positions, i.e. ~100 haplotypes, in one block) is worth building.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather_pairs.hip -o /tmp/ubench_pairs && /tmp/ubench_pairs
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) gather(const uint4 *__restrict__ tab, uint64_t n_blocks, int iters, uint64_t *__restrict__ out) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const uint64_t r = x * 0x2545F4914F6CDD1Dull;
        const uint4 *p = tab + (r % n_blocks) * 16; // a 256-byte block
        const uint4 *q = MODE == 2 ? tab + ((r >> 20) % n_blocks) * 16 : p + 8;
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) { const uint4 v = p[(k * 3) & 7]; s += v.x ^ v.y ^ v.z ^ v.w; }
        if (MODE >= 1) {
#pragma unroll
            for (int k = 0; k < 5; k++) { const uint4 v = q[(k * 3) & 7]; s += v.x ^ v.y ^ v.z ^ v.w; }
        }
        acc += s;
        x += s;
    }
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 300, wps = 5, grid = cus * wps;
    uint64_t *out = nullptr;
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (size_t mb : {1280, 2560}) {
        uint4 *tab = nullptr;
        CHECK(hipMalloc(&tab, mb << 20));
        CHECK(hipMemset(tab, 1, mb << 20));
        const uint64_t n_blocks = (uint64_t)(mb << 20) / 256;
        const char *what[3] = {"one line", "two adjacent lines (one 256-byte block)", "two independent lines"};
        for (int mode = 0; mode < 3; mode++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(a, 0));
                if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(grid), dim3(256), 0, 0, tab, n_blocks, iters, out);
                else if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(grid), dim3(256), 0, 0, tab, n_blocks, iters, out);
                else hipLaunchKernelGGL(gather<2>, dim3(grid), dim3(256), 0, 0, tab, n_blocks, iters, out);
                CHECK(hipEventRecord(b, 0));
                CHECK(hipEventSynchronize(b));
                CHECK(hipEventElapsedTime(&ms, a, b));
            }
            printf("table %5zu MB, %-42s per step: %6.2f G steps/s (%.3f ms)\n", mb, what[mode], (double)grid * 256 * iters / ms / 1e6, ms);
        }
        CHECK(hipFree(tab));
    }
    return 0;
}

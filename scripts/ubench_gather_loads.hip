// ubench_gather_loads.hip -- dependent random gathers of one 128-byte line, reading L = 1, 2, 5 or 8 of its 16-byte pieces, for tables from
// 853 MB to 32 GB: what a rank probe costs as a function of the image's footprint and of how much of the line it reads.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather_loads.hip -o /tmp/ubench_loads && /tmp/ubench_loads
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int L>
__global__ void __launch_bounds__(256) gather(const uint4 *__restrict__ tab, uint64_t n_lines, int iters, uint64_t *__restrict__ out) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const uint4 *p = tab + ((x * 0x2545F4914F6CDD1Dull) % n_lines) * 8;
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < L; k++) { const uint4 v = p[(k * 3) & 7]; s += v.x ^ v.y ^ v.z ^ v.w; }
        acc += s;
        x += s;
    }
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 300, wps = 5, grid = cus * wps;
    uint64_t *out = nullptr;
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const size_t sizes_mb[] = {853, 2048, 3072, 4096, 6144, 16384, 32768};
    for (size_t mb : sizes_mb) {
        uint4 *tab = nullptr;
        CHECK(hipMalloc(&tab, mb << 20));
        CHECK(hipMemset(tab, 1, mb << 20));
        const uint64_t n_lines = (uint64_t)(mb << 20) / 128;
        for (int L : {1, 2, 5, 8}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(a, 0));
                if (L == 1) hipLaunchKernelGGL(gather<1>, dim3(grid), dim3(256), 0, 0, tab, n_lines, iters, out);
                else if (L == 2) hipLaunchKernelGGL(gather<2>, dim3(grid), dim3(256), 0, 0, tab, n_lines, iters, out);
                else if (L == 5) hipLaunchKernelGGL(gather<5>, dim3(grid), dim3(256), 0, 0, tab, n_lines, iters, out);
                else hipLaunchKernelGGL(gather<8>, dim3(grid), dim3(256), 0, 0, tab, n_lines, iters, out);
                CHECK(hipEventRecord(b, 0));
                CHECK(hipEventSynchronize(b));
                CHECK(hipEventElapsedTime(&ms, a, b));
            }
            printf("table %6zu MB, %d x 16 B of a line: %7.2f G lines/s (%.3f ms)\n", mb, L, (double)grid * 256 * iters / ms / 1e6, ms);
        }
        CHECK(hipFree(tab));
    }
    return 0;
}

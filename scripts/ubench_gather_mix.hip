// ubench_gather_mix.hip -- the find_mems access mix at chr22 scale: most dependent gathers go to a table within TLB reach (the 853 MB PAIRS
// image), one in N to a table far beyond it (the 16 GiB seed table).  Does the minority of far gathers slow the whole stream down?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather_mix.hip -o /tmp/ubench_mix && /tmp/ubench_mix
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every `every`-th iteration of a lane reads one 16-byte entry of the far table instead of a 128-byte record of the near one (every = 0: never)
__global__ void __launch_bounds__(256) gather(const uint4 *__restrict__ near_tab, uint64_t n_near, const uint4 *__restrict__ far_tab, uint64_t n_far, int every,
                                              int iters, uint64_t *__restrict__ out) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const uint64_t h = x * 0x2545F4914F6CDD1Dull;
        uint32_t s = 0;
        if (every && ((it + (int)(threadIdx.x & 15)) % every) == 0) {
            const uint4 v = far_tab[h % n_far];
            s = v.x ^ v.y ^ v.z ^ v.w;
        } else {
            const uint4 *p = near_tab + (h % n_near) * 8;
#pragma unroll
            for (int k = 0; k < 5; k++) { const uint4 v = p[k + 3]; s += v.x ^ v.y ^ v.z ^ v.w; } // five 16-byte loads of one line, like a PAIRS probe
        }
        acc += s;
        x += s;
    }
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t near_mb = argc > 1 ? (size_t)atoll(argv[1]) : 853;
    const int iters = 400, wps = 5;
    uint4 *near_tab = nullptr;
    CHECK(hipMalloc(&near_tab, near_mb << 20));
    CHECK(hipMemset(near_tab, 1, near_mb << 20));
    const size_t far_sizes_mb[] = {256, 1024, 4096, 16384};
    for (size_t far_mb : far_sizes_mb) {
        uint4 *far_tab = nullptr;
        CHECK(hipMalloc(&far_tab, far_mb << 20));
        CHECK(hipMemset(far_tab, 1, far_mb << 20));
        for (int every : {0, 20, 10, 5, 1}) {
            const int grid = cus * wps;
            uint64_t *out = nullptr;
            CHECK(hipMalloc(&out, (size_t)grid * 256 * 8));
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(a, 0));
                hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, 0, near_tab, (uint64_t)(near_mb << 20) / 128, far_tab, (uint64_t)(far_mb << 20) / 16, every, iters, out);
                CHECK(hipEventRecord(b, 0));
                CHECK(hipEventSynchronize(b));
                CHECK(hipEventElapsedTime(&ms, a, b));
            }
            const double acc = (double)grid * 256 * iters;
            printf("near %5zu MB, far %6zu MB, one far gather in %2d: %7.2f G gathers/s (%.3f ms)\n", near_mb, far_mb, every, acc / ms / 1e6, ms);
            CHECK(hipFree(out));
        }
        CHECK(hipFree(far_tab));
    }
    return 0;
}

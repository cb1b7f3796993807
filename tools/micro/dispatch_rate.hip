// tools/micro/dispatch_rate.hip -- how fast can the chip START workgroups?  An (almost) empty kernel on the trace kernel's grid:
// 65 536 x 64 threads with 5 KB of LDS each, then the same threads as 128- and 256-thread workgroups and without LDS; and a
// kernel that idles for a fixed number of clocks, to see how long a freed wave slot stays empty.
// hipcc --offload-arch=gfx950 -O3 tools/micro/dispatch_rate.hip -o /tmp/dispatch_rate && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_empty(unsigned* out, int spin, unsigned* rec = nullptr, int nrec = 0) {
    extern __shared__ unsigned s[];
    s[threadIdx.x] = threadIdx.x;
    if (spin > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) __builtin_amdgcn_s_sleep(8);
    }
    if (out && s[threadIdx.x ^ 1] == 0xFFFFFFFFu) out[blockIdx.x] = 1;
    // the trace kernel's epilogue: nrec coalesced 4-byte stores per lane (structure of arrays), then the wave ends
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < nrec; ++r) rec[r * n + gid] = (unsigned)gid + r;
}

static unsigned* g_rec = nullptr;
static float time_launch(int wgs, int threads, int lds, int spin, int reps, int nrec = 0) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ms;
    for (int r = 0; r < reps + 2; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(threads), lds, 0, (unsigned*)nullptr, spin, g_rec, nrec);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2] * 1e3f;
}

int main() {
    const int total = 65536 * 64;
    printf("threads/WG  LDS/WG  spin clk   us per launch   WG/us   waves/us\n");
    for (int spin : {0, 20000, 80000}) {
        for (int threads : {64, 128, 256}) {
            for (int ldsw : {0, 5120}) {
                const int wgs = total / threads, lds = ldsw * (threads / 64);
                const float us = time_launch(wgs, threads, lds, spin, 9);
                printf("%9d %7d %9d %15.1f %7.1f %10.1f\n", threads, lds, spin, us, wgs / us, (total / 64) / us);
            }
        }
    }
    hipMalloc(&g_rec, (size_t)total * 4 * 9);
    printf("with 9 coalesced dword stores per lane before the wave ends (64-thread workgroups, 5 KB LDS):\n");
    for (int spin : {0, 20000, 80000}) {
        const float a = time_launch(65536, 64, 5120, spin, 9, 0), b = time_launch(65536, 64, 5120, spin, 9, 9);
        printf("  spin %6d: %8.1f us without stores, %8.1f us with\n", spin, a, b);
    }
    return 0;
}

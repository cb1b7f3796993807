// tools/micro/queue_share.hip -- does a SMALL kernel on one stream get wave slots while a LARGE kernel of another stream is still
// handing out workgroups?  (The compaction kernels of one scan chain beside the trace launch of the other chain.)
// Kernel A: 65 536 one-wave workgroups, 5 KB LDS, 64 VGPRs (the trace kernel's shape: 32 of them fill a CU), each spinning ~35 us.
// Kernel S: 64 one-wave workgroups (or 16 x 256 threads), submitted on another stream ~60 us after A; per workgroup start / end
// stamps (s_memrealtime, 100 MHz).  Printed per stream pair: when S's first / last workgroup started and when S ended, against
// the moment A handed out its last workgroup.
// hipcc --offload-arch=gfx950 -O3 tools/micro/queue_share.hip -o build_variants/queue_share
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <unistd.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool FAT>
__global__ __launch_bounds__(1024) void k_spin(unsigned long long* stamps, int clk) {
    extern __shared__ unsigned s[];
    if (FAT) asm volatile("v_mov_b32 v63, 0" ::: "v63");      // 64 VGPRs: eight waves fill a SIMD's register file
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < clk) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t_start;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main() {
    const int NS = 6;
    hipStream_t st[NS];
    for (int i = 0; i < NS; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    hipStream_t hi;
    int lo_p = 0, hi_p = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, hi_p));
    printf("stream priority range: least %d .. greatest %d\n", lo_p, hi_p);
    const int wgsA = 65536, clkA = 70000, wgsS = 64;
    unsigned long long *sa, *ss;
    CK(hipMalloc(&sa, (size_t)wgsA * 16));
    CK(hipMalloc(&ss, (size_t)4096 * 16));
    std::vector<unsigned long long> ha(2 * wgsA), hs(2 * 4096);
    auto run = [&](const char* what, hipStream_t a, hipStream_t b, int s_threads, int s_lds, int s_wgs, int delay_us = 20) {
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_spin<true>, dim3(wgsA), dim3(64), 5120, a, sa, clkA);
        usleep(delay_us);                              // A is in the middle of handing out its workgroups
        hipLaunchKernelGGL(k_spin<false>, dim3(s_wgs), dim3(s_threads), s_lds, b, ss, 2000);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ha.data(), sa, (size_t)wgsA * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), ss, (size_t)s_wgs * 16, hipMemcpyDeviceToHost));
        unsigned long long a0 = ~0ull, a_last = 0, a_end = 0, s0 = ~0ull, s_last = 0, s_end = 0;
        for (int i = 0; i < wgsA; ++i) { a0 = std::min(a0, ha[2 * i]); a_last = std::max(a_last, ha[2 * i]); a_end = std::max(a_end, ha[2 * i + 1]); }
        for (int i = 0; i < s_wgs; ++i) { s0 = std::min(s0, hs[2 * i]); s_last = std::max(s_last, hs[2 * i]); s_end = std::max(s_end, hs[2 * i + 1]); }
        auto us = [&](unsigned long long t) { return (double)((long long)(t - a0)) / 100.0; };
        printf("%-52s A hands out its last WG at %6.1f us, ends %6.1f | S first WG %6.1f, last WG %6.1f, ends %6.1f\n", what,
               us(a_last), us(a_end), us(s0), us(s_last), us(s_end));
        return 0;
    };
    // the chain as the scan pipeline has it: stream b runs a LARGE kernel and then, behind it in the same stream, the small one,
    // while stream a is handing out the workgroups of another large kernel
    unsigned long long* sb;
    CK(hipMalloc(&sb, (size_t)wgsA * 16));
    std::vector<unsigned long long> hb(2 * wgsA);
    auto chain = [&](const char* what, hipStream_t a, hipStream_t b, int s_threads, int s_wgs) {
        CK(hipDeviceSynchronize());
        const int wgsB = 16384;
        hipLaunchKernelGGL(k_spin<true>, dim3(wgsB), dim3(64), 5120, b, sb, clkA);
        hipLaunchKernelGGL(k_spin<false>, dim3(s_wgs), dim3(s_threads), 0, b, ss, 2000);
        hipLaunchKernelGGL(k_spin<true>, dim3(wgsA), dim3(64), 5120, a, sa, clkA);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ha.data(), sa, (size_t)wgsA * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), sb, (size_t)wgsB * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), ss, (size_t)s_wgs * 16, hipMemcpyDeviceToHost));
        unsigned long long a0 = ~0ull, a_last = 0, a_end = 0, b_last = 0, b_end = 0, s0 = ~0ull, s_last = 0, s_end = 0, t0 = ~0ull;
        for (int i = 0; i < wgsA; ++i) { a0 = std::min(a0, ha[2 * i]); a_last = std::max(a_last, ha[2 * i]); a_end = std::max(a_end, ha[2 * i + 1]); }
        for (int i = 0; i < wgsB; ++i) { t0 = std::min(t0, hb[2 * i]); b_last = std::max(b_last, hb[2 * i]); b_end = std::max(b_end, hb[2 * i + 1]); }
        for (int i = 0; i < s_wgs; ++i) { s0 = std::min(s0, hs[2 * i]); s_last = std::max(s_last, hs[2 * i]); s_end = std::max(s_end, hs[2 * i + 1]); }
        t0 = std::min(t0, a0);
        auto us = [&](unsigned long long t) { return (double)((long long)(t - t0)) / 100.0; };
        printf("%-40s B (same stream as S) last WG %6.1f ends %6.1f | A first WG %6.1f last WG %6.1f ends %6.1f | S first WG %6.1f, last WG %6.1f, ends %6.1f\n",
               what, us(b_last), us(b_end), us(a0), us(a_last), us(a_end), us(s0), us(s_last), us(s_end));
        return 0;
    };
    char label[128];
    for (int rep = 0; rep < 2; ++rep) {
        if (chain("chain: S = 32 x 64 threads", st[0], st[1], 64, 32)) return 1;
        if (chain("chain: S = 8192 x 256 threads", st[0], st[1], 256, 4096)) return 1;
        if (chain("chain: S = 32 x 64, S on HIGH PRIORITY", st[0], hi, 64, 32)) return 1;
        if (chain("chain: S = 4096 x 256, HIGH PRIORITY", st[0], hi, 256, 4096)) return 1;
        for (int j = 1; j < NS; ++j) {
            snprintf(label, sizeof label, "S = 64 x 64 threads, streams 0 / %d", j);
            if (run(label, st[0], st[j], 64, 0, wgsS)) return 1;
        }
        if (run("S = 64 x 64 threads, 5 KB LDS, streams 0 / 1", st[0], st[1], 64, 5120, wgsS)) return 1;
        if (run("S = 16 x 256 threads, streams 0 / 1", st[0], st[1], 256, 0, 16)) return 1;
        if (run("S = 4096 x 256 threads, streams 0 / 1", st[0], st[1], 256, 0, 4096)) return 1;
        if (run("S = 4096 x 64 threads, streams 0 / 1", st[0], st[1], 64, 0, 4096)) return 1;
        if (run("S = 64 x 64 threads on a HIGH PRIORITY stream", st[0], hi, 64, 0, wgsS)) return 1;
        if (run("S = 4096 x 256 threads on a HIGH PRIORITY stream", st[0], hi, 256, 0, 4096)) return 1;
        if (run("S = 16 x 1024 threads on a HIGH PRIORITY stream", st[0], hi, 1024, 0, 16)) return 1;
    }
    return 0;
}

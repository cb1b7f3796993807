// tools/micro/wait_value.hip -- can a stream be made to start its next kernel the moment another stream's kernel has
// DISPATCHED its last workgroup (not when it has finished)?  Kernel A's last workgroup writes a sequence number to a signal
// word at its first instruction; stream 1 holds kernel B behind hipStreamWaitValue64(word >= seq).  Every workgroup stamps
// s_memrealtime (100 MHz) at start and end, so the overlap can be read off: B's first start against A's last start / A's end.
// hipcc --offload-arch=gfx950 -O3 tools/micro/wait_value.hip -o /tmp/wait_value && /tmp/wait_value
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <unistd.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// spins for `clk` shader clocks; every 64th workgroup four times as long (a tail)
__global__ void k_work(unsigned long long* stamps, int clk, unsigned long long* flag, unsigned long long seq) {
    extern __shared__ unsigned s[];
    s[threadIdx.x] = threadIdx.x;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    if (flag && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long want = (blockIdx.x % 64 == 63) ? 4ll * clk : clk;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < want) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t_start;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (s[threadIdx.x ^ 1] == 0xFFFFFFFFu) stamps[0] = 0;
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned long long* flag = nullptr;
    CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    printf("signal word at %p\n", (void*)flag);
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const int wgs = 32768, clk = 60000;       // ~ 8192 resident waves -> 4 rounds of ~28 us + the tail of long workgroups
    unsigned long long *sa, *sb;
    CK(hipMalloc(&sa, (size_t)wgs * 16));
    CK(hipMalloc(&sb, (size_t)wgs * 16));
    std::vector<unsigned long long> ha(2 * wgs), hb(2 * wgs);
    auto report = [&](const char* what) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ha.data(), sa, (size_t)wgs * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), sb, (size_t)wgs * 16, hipMemcpyDeviceToHost));
        unsigned long long a0 = ~0ull, a_last_start = 0, a_end = 0, b0 = ~0ull, b_end = 0;
        for (int i = 0; i < wgs; ++i) {
            a0 = std::min(a0, ha[2 * i]); a_last_start = std::max(a_last_start, ha[2 * i]); a_end = std::max(a_end, ha[2 * i + 1]);
            b0 = std::min(b0, hb[2 * i]); b_end = std::max(b_end, hb[2 * i + 1]);
        }
        auto us = [&](unsigned long long t) { return (double)((long long)(t - a0)) / 100.0; };
        printf("%-44s A: last WG starts %7.1f us, ends %7.1f us | B: first WG starts %7.1f us, ends %7.1f us\n", what,
               us(a_last_start), us(a_end), us(b0), us(b_end));
        return 0;
    };
    for (int rep = 0; rep < 2; ++rep) {
        // 1. B behind an event recorded after A (ordinary stream dependency): B starts when A has ENDED
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipLaunchKernelGGL(k_work, dim3(wgs), dim3(64), 5120, s0, sa, clk, (unsigned long long*)nullptr, 0ull);
        CK(hipEventRecord(ev, s0));
        CK(hipStreamWaitEvent(s1, ev, 0));
        hipLaunchKernelGGL(k_work, dim3(wgs), dim3(64), 5120, s1, sb, clk, (unsigned long long*)nullptr, 0ull);
        if (report("B after A's completion event:")) return 1;
        // 2. B with no dependency: both dispatch at once
        hipLaunchKernelGGL(k_work, dim3(wgs), dim3(64), 5120, s0, sa, clk, (unsigned long long*)nullptr, 0ull);
        hipLaunchKernelGGL(k_work, dim3(wgs), dim3(64), 5120, s1, sb, clk, (unsigned long long*)nullptr, 0ull);
        if (report("B with no dependency:")) return 1;
        // 3. B behind the wait for A's last workgroup to START
        const unsigned long long seq = 100 + rep;
        CK(hipStreamWaitValue64(s1, flag, seq, hipStreamWaitValueGte, ~0ull));
        hipLaunchKernelGGL(k_work, dim3(wgs), dim3(64), 5120, s1, sb, clk, (unsigned long long*)nullptr, 0ull);
        hipLaunchKernelGGL(k_work, dim3(wgs), dim3(64), 5120, s0, sa, clk, flag, seq);
        // safety net: if the wait never fires, release it from the host side instead of hanging the box
        bool released = false;
        for (int spin = 0; spin < 3000; ++spin) {
            if (hipStreamQuery(s1) == hipSuccess) { released = true; break; }
            usleep(1000);
        }
        if (!released) {
            printf("the wait did not fire within 3 s: releasing it with hipStreamWriteValue64\n");
            CK(hipStreamWriteValue64(s0, flag, seq, 0));
        }
        if (report("B behind hipStreamWaitValue64(A's last WG):")) return 1;
    }
    return 0;
}

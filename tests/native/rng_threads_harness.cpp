// tests/native/rng_threads_harness.cpp -- the streaming path of csrc/lrc_nprandom.cpp (one generating thread, flag / transform
// workers, the caller walking the counts) under ThreadSanitizer or AddressSanitizer: the same draws and the same generator
// state as the sequential path (threads < 0), call after call, for several thread counts.  Built by
// tests/test_native_sanitizers.py with -DLRC_TSAN (condition-variable waits without a time-out: the sanitizer runtime of this
// toolchain does not know pthread_cond_clockwait) and with the run-time dispatch of the AVX2 clones removed.
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/lidarcast.h"

extern "C" int lrc_internal_fail(int c, const char* m) { std::fprintf(stderr, "fail %d %s\n", c, m); return c; }
extern "C" void lrc_internal_set_thread_budget(long budget);      // test hook: later thread starts fail like EAGAIN

int main() {
    lrc_mt19937_state a{}, b{};
    for (int i = 0; i < 624; ++i) a.key[i] = 1812433253u * (unsigned)i + 12345u;
    a.pos = 624;
    b = a;
    const unsigned long P = 9, nn = 40000, nu = 20000;
    std::vector<double> z1(P * nn), u1(P * nu), z2(P * nn), u2(P * nu);
    for (int rep = 0; rep < 6; ++rep) {
        const int t = 3 + rep % 4;
        if (lrc_rng_scan_draws(&a, P, nn, nu, 0.5, 2.0, z1.data(), u1.data(), t)) return 1;
        if (lrc_rng_scan_draws(&b, P, nn, nu, 0.5, 2.0, z2.data(), u2.data(), -t)) return 1;
        if (std::memcmp(z1.data(), z2.data(), z1.size() * 8) || std::memcmp(u1.data(), u2.data(), u1.size() * 8) ||
            std::memcmp(&a, &b, sizeof(a))) { std::puts("MISMATCH"); return 2; }
    }
    // thread starts that fail: no generator (sequential path takes over), a generator and no / some workers, a sequential pool
    // that stays empty or short -- same draws, same state, nothing left joinable (the sanitizers watch the wind-down)
    for (long budget = 0; budget < 5; ++budget) {
        for (int t : {6, -4}) {
            lrc_internal_set_thread_budget(budget);
            const int rc = lrc_rng_scan_draws(&a, P, nn, nu, 0.5, 2.0, z1.data(), u1.data(), t);
            lrc_internal_set_thread_budget(-1);
            if (rc) return 3;
            if (lrc_rng_scan_draws(&b, P, nn, nu, 0.5, 2.0, z2.data(), u2.data(), -3)) return 1;
            if (std::memcmp(z1.data(), z2.data(), z1.size() * 8) || std::memcmp(u1.data(), u2.data(), u1.size() * 8) ||
                std::memcmp(&a, &b, sizeof(a))) { std::puts("MISMATCH after failed thread starts"); return 4; }
        }
    }
    std::puts("rng threads harness ok");
    return 0;
}

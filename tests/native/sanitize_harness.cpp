// Sanitizer harness (CPU only; GPU AddressSanitizer is not available on the pool): the host BVH builder of the product
// (csrc/bvh_build.cpp, threaded) and the C oracle, on random, snapped, single-point and tiny meshes, under
// -fsanitize=address,undefined.  Built and run by tests/test_native_sanitizers.py.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
#include "lrc_bvh.h"
extern "C" {
void orc_cast_brute(const float*, const uint32_t*, uint64_t, const float*, uint64_t, float*, uint32_t*);
struct orc_bvh; orc_bvh* orc_bvh_build(const float*, const uint32_t*, uint64_t); void orc_bvh_free(orc_bvh*);
void orc_cast_bvh(const orc_bvh*, const float*, uint64_t, float*, uint32_t*, int);
void orc_cast_bvh_diag(const orc_bvh*, const float*, uint64_t, float*, uint32_t*, uint32_t*, int);
void orc_witness_f64(const orc_bvh*, const float*, uint64_t, double*, uint32_t*, double*, int);
void orc_witness_tri_f64(const float*, const uint32_t*, const float*, const uint32_t*, uint64_t, double*, double*);
}
static float rnd() { return (float)rand() / RAND_MAX; }
int main() {
    for (int round = 0; round < 6; ++round) {
        uint64_t T = round == 0 ? 1 : round == 1 ? 5 : round == 2 ? 1000 : round == 3 ? 50000 : round == 4 ? 7 : 200000;
        std::vector<float> v(T * 9);
        std::vector<uint32_t> f(T * 3);
        for (uint64_t i = 0; i < T * 9; ++i) v[i] = (round == 4) ? 0.5f : (round % 2 ? std::floor(rnd() * 8) / 4 : rnd() * 4);
        for (uint64_t i = 0; i < T * 3; ++i) f[i] = (uint32_t)i;
        lrc::HostBVH h; lrc::BuildOptions opt; opt.threads = 4;
        lrc::build_bvh(v.data(), T * 3, f.data(), T, nullptr, nullptr, opt, &h);
        for (int threads : {1, 16}) {     // the layout must not depend on how many tasks built it (every rank of a
            lrc::HostBVH g; lrc::BuildOptions o2; o2.threads = threads;      // multi-GPU job builds its own replica)
            lrc::build_bvh(v.data(), T * 3, f.data(), T, nullptr, nullptr, o2, &g);
            auto same = [](const std::vector<float>& a, const std::vector<float>& b) {   // bytes: child refs are ints
                return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
            };
            if (!same(g.nodes, h.nodes) || !same(g.tri_rec, h.tri_rec) || g.slot_prim != h.slot_prim ||
                g.max_depth != h.max_depth) {
                printf("NONDETERMINISTIC round %d threads %d\n", round, threads);
                return 2;
            }
        }
        uint64_t N = 2000;
        std::vector<float> rays(N * 6), t1(N), t2(N); std::vector<uint32_t> p1(N), p2(N);
        for (uint64_t i = 0; i < N * 6; ++i) rays[i] = rnd() * 4 - (i % 6 >= 3 ? 2 : 0);
        orc_bvh* b = orc_bvh_build(v.data(), f.data(), T);
        orc_cast_bvh(b, rays.data(), N, t2.data(), p2.data(), 3);
        if (T <= 50000) {
            orc_cast_brute(v.data(), f.data(), T, rays.data(), N, t1.data(), p1.data());
            for (uint64_t i = 0; i < N; ++i) if (p1[i] != p2[i]) { printf("MISMATCH round %d ray %lu\n", round, (unsigned long)i); return 1; }
        }
        {   // the diagnostic cast returns the same hits; the float64 witness and its one-triangle form run clean,
            // also on rays with non-finite components (finite-ray contract: a miss)
            std::vector<float> t3(N); std::vector<uint32_t> p3(N), rej(N), pw(N);
            std::vector<double> tw(N), mw(N), tt(N), mt(N);
            orc_cast_bvh_diag(b, rays.data(), N, t3.data(), p3.data(), rej.data(), 3);
            for (uint64_t i = 0; i < N; ++i) if (p3[i] != p2[i]) { printf("MISMATCH diag round %d ray %lu\n", round, (unsigned long)i); return 1; }
            rays[3] = NAN; rays[6 * 7 + 1] = INFINITY;
            orc_witness_f64(b, rays.data(), N, tw.data(), pw.data(), mw.data(), 3);
            orc_witness_tri_f64(v.data(), f.data(), rays.data(), pw.data(), N, tt.data(), mt.data());
            if (!std::isinf(tw[0]) || !std::isinf(tw[7])) { printf("MISMATCH non-finite ray hit, round %d\n", round); return 1; }
            for (uint64_t i = 0; i < N; ++i)
                if (pw[i] != 0xFFFFFFFFu && tt[i] != tw[i]) { printf("MISMATCH witness round %d ray %lu\n", round, (unsigned long)i); return 1; }
        }
        orc_bvh_free(b);
        printf("round %d T=%lu nodes=%lu depth=%u ok\n", round, (unsigned long)T, (unsigned long)h.num_nodes, h.max_depth);
    }
    return 0;
}

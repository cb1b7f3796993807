// Sanitizer harness (CPU only; GPU AddressSanitizer is not available on the pool): the host BVH builder of the product
// (csrc/bvh_build.cpp, threaded) and the C oracle, on random, snapped, single-point and tiny meshes, under
// -fsanitize=address,undefined.  Built and run by tests/test_native_sanitizers.py.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
#include <algorithm>
#include "lrc_bvh.h"
#include "lrc_qnodes.h"
extern "C" {
void orc_cast_brute(const float*, const uint32_t*, uint64_t, const float*, uint64_t, float*, uint32_t*);
struct orc_bvh; orc_bvh* orc_bvh_build(const float*, const uint32_t*, uint64_t); void orc_bvh_free(orc_bvh*);
void orc_cast_bvh(const orc_bvh*, const float*, uint64_t, float*, uint32_t*, int);
void orc_cast_bvh_diag(const orc_bvh*, const float*, uint64_t, float*, uint32_t*, uint32_t*, int);
void orc_witness_f64(const orc_bvh*, const float*, uint64_t, double*, uint32_t*, double*, int);
void orc_witness_tri_f64(const float*, const uint32_t*, const float*, const uint32_t*, uint64_t, double*, double*);
}
static float rnd() { return (float)rand() / RAND_MAX; }

// ---- quantised node images (csrc/lrc_qnodes.cpp): host-side checks, and the kernel's box arithmetic restated --------
static float qdec(uint32_t q) { uint32_t u = 0x40000000u | (q << 8); float f; std::memcpy(&f, &u, 4); return f; }
static float safe_inv(float d) { float a = std::fabs(d); float s = a < 1e-30f ? std::copysign(1e-30f, d) : d; return 1.0f / s; }
static int32_t iref(float f) { int32_t r; std::memcpy(&r, &f, 4); return r; }
// the padded interval of the ray against a box given in NORMALISED coordinates, as trace_kernel<QN> forms it
// (lrc_device.h slab_interval on the normalised slab constants); compiled with -ffp-contract=off, fmaf explicit
static void qinterval(const float n_lo[3], const float n_hi[3], const float ixq[3], const float oxq[3], float& tn, float& tf) {
    float nn = 0.0f, ff = INFINITY;
    for (int a = 0; a < 3; ++a) {
        const float t0 = std::fmaf(n_lo[a], ixq[a], -oxq[a]), t1 = std::fmaf(n_hi[a], ixq[a], -oxq[a]);
        nn = std::fmax(nn, std::fmin(t0, t1));
        ff = std::fmin(ff, std::fmax(t0, t1));
    }
    tn = std::fmaf(nn, 0.999755859375f, -1.52587890625e-05f);
    tf = std::fmaf(ff, 1.000244140625f, 1.52587890625e-05f);
}
// 0: fine.  Checks: every grid box holds its float32 box plus the margin; the float32 image is the decode of the packed
// one; the four-wide collapse reaches exactly the leaves of the binary tree with the same boxes; and, for every ray with
// a brute-force hit, every box on the path from the root to the hit triangle's leaf passes the kernel's test in
// normalised coordinates (tn <= tf and tn <= t) -- i.e. the quantised traversal cannot cull the definition's hit.
static int check_qnodes(const lrc::HostBVH& h, const float* rays, uint64_t N, const float* t_bf, const uint32_t* p_bf, int round) {
    float base[3], W[3], invW[3];
    lrc::QGrid g;
    if (!lrc::make_qgrid(h, base, W, invW, g)) { printf("round %d: no grid for this scene\n", round); return 0; }
    std::vector<uint32_t> q8, q16; std::vector<float> n16, n32; double infl = 0; uint64_t num4 = 0;
    if (!lrc::build_qnodes(h, g, q8, n16, &infl) || !lrc::build_q4nodes(h, g, q16, n32, &num4)) { printf("QNODES build failed round %d\n", round); return 3; }
    const double m = lrc::kQnodeMargin;
    for (uint64_t i = 0; i < h.num_nodes; ++i)
        for (int c = 0; c < 2; ++c)
            for (int a = 0; a < 3; ++a) {
                const uint32_t w = q8[i * 8 + c * 4 + a], ql = w & 0xFFFFu, qh = w >> 16;
                if (iref(h.nodes[i * 16 + 12 + c]) == ~0) { if (ql != 32767u || qh != 0u) { printf("QNODES empty child round %d\n", round); return 3; } continue; }
                const double cell = g.Wd[a] / 16384.0;
                const double xl = g.bd[a] + (2.0 + ql / 16384.0) * g.Wd[a], xh = g.bd[a] + (2.0 + qh / 16384.0) * g.Wd[a];
                const double lo = h.nodes[i * 16 + c * 6 + a], hi = h.nodes[i * 16 + c * 6 + 3 + a];
                if (qh > 32767u || !(xl <= lo - m * cell * 0.999) || !(xh >= hi + m * cell * 0.999) || !(xl > lo - (1.0 + 2 * m) * cell) || !(xh < hi + (1.0 + 2 * m) * cell)) {
                    printf("QNODES containment round %d node %lu\n", round, (unsigned long)i); return 3; }
                if (n16[i * 16 + c * 6 + a] != qdec(ql) || n16[i * 16 + c * 6 + 3 + a] != qdec(qh)) { printf("QNODES decode round %d\n", round); return 3; }
            }
    // leaves reachable through the four-wide nodes == leaves of the binary tree (as multisets of references)
    std::vector<int32_t> l2, l4, st;
    st.push_back(0);
    while (!st.empty()) { int32_t n = st.back(); st.pop_back();
        for (int c = 0; c < 2; ++c) { int32_t r = iref(h.nodes[(size_t)n * 16 + 12 + c]); if (r >= 0) st.push_back(r); else if (r != ~0) l2.push_back(r); } }
    st.push_back(0);
    while (!st.empty()) { int32_t n = st.back(); st.pop_back();
        if ((uint64_t)n >= num4) { printf("QNODES wide index round %d\n", round); return 3; }
        for (int k = 0; k < 4; ++k) { int32_t r; std::memcpy(&r, &q16[(size_t)n * 16 + k * 4 + 3], 4);
            if (r != iref(n32[(size_t)n * 32 + k * 8 + 6])) { printf("QNODES wide refs differ round %d\n", round); return 3; }
            if (r >= 0) st.push_back(r); else if (r != ~0) l4.push_back(r); } }
    std::sort(l2.begin(), l2.end()); std::sort(l4.begin(), l4.end());
    if (l2 != l4) { printf("QNODES wide leaves differ round %d (%lu vs %lu)\n", round, (unsigned long)l2.size(), (unsigned long)l4.size()); return 3; }
    // parents, and the leaf of every triangle row
    std::vector<int32_t> parent(h.num_nodes, -1), pchild(h.num_nodes, 0);
    std::vector<int32_t> leaf_node(h.slot_prim.size(), -1), leaf_child(h.slot_prim.size(), 0), slot_of_prim(h.slot_prim.size(), -1);
    for (uint64_t i = 0; i < h.num_nodes; ++i)
        for (int c = 0; c < 2; ++c) { int32_t r = iref(h.nodes[i * 16 + 12 + c]);
            if (r >= 0) { parent[r] = (int32_t)i; pchild[r] = c; }
            else if (r != ~0) { uint32_t enc = (uint32_t)~r; for (uint32_t k = 0; k < (enc & 7u); ++k) { leaf_node[(enc >> 3) + k] = (int32_t)i; leaf_child[(enc >> 3) + k] = c; } } }
    for (size_t sidx = 0; sidx < h.slot_prim.size(); ++sidx) if (h.slot_prim[sidx] < slot_of_prim.size()) slot_of_prim[h.slot_prim[sidx]] = (int32_t)sidx;
    uint64_t checked = 0;
    for (uint64_t i = 0; i < N; ++i) {
        if (!(t_bf[i] < INFINITY)) continue;
        const float* r = rays + i * 6;
        float ixq[3], oxq[3]; bool near = true;
        for (int a = 0; a < 3; ++a) {
            ixq[a] = safe_inv(r[3 + a]) * W[a];
            oxq[a] = ((r[a] - base[a]) * invW[a]) * ixq[a];
            near = near && std::fabs(r[a] - base[a]) <= lrc::kQnodeNearBase * W[a] && std::fabs(r[a]) <= lrc::kQnodeNearOrigin * W[a] && std::fabs(r[3 + a]) <= 0x1p60f;
        }
        if (!near) continue;                      // the kernel walks the float32 nodes for this ray
        const int32_t slot = slot_of_prim[p_bf[i]];
        int32_t n = leaf_node[slot], c = leaf_child[slot];
        while (n >= 0) {
            float tn, tf;
            qinterval(&n16[(size_t)n * 16 + c * 6], &n16[(size_t)n * 16 + c * 6 + 3], ixq, oxq, tn, tf);
            if (!(tn <= tf) || !(tn <= t_bf[i])) { printf("QNODES would cull the hit: round %d ray %lu node %d child %d tn %g tf %g t %g\n", round, (unsigned long)i, n, c, tn, tf, t_bf[i]); return 3; }
            c = pchild[n]; n = parent[n];
        }
        ++checked;
    }
    printf("round %d: grid W = %g %g %g, leaf inflation %.3f, %lu hit rays checked along their paths, %lu four-wide nodes\n", round, W[0], W[1], W[2], infl,
           (unsigned long)checked, (unsigned long)num4);
    return 0;
}
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
        uint64_t N = T <= 1000 ? 20000 : 2000;
        std::vector<float> rays(N * 6), t1(N), t2(N); std::vector<uint32_t> p1(N), p2(N);
        for (uint64_t i = 0; i < N * 6; ++i) rays[i] = rnd() * 4 - (i % 6 >= 3 ? 2 : 0);
        orc_bvh* b = orc_bvh_build(v.data(), f.data(), T);
        orc_cast_bvh(b, rays.data(), N, t2.data(), p2.data(), 3);
        if (T <= 50000) {
            orc_cast_brute(v.data(), f.data(), T, rays.data(), N, t1.data(), p1.data());
            for (uint64_t i = 0; i < N; ++i) if (p1[i] != p2[i]) { printf("MISMATCH round %d ray %lu\n", round, (unsigned long)i); return 1; }
            if (int rc = check_qnodes(h, rays.data(), N, t1.data(), p1.data(), round)) return rc;
            if (T <= 1000) {     // and out to the edge of the bound the margin is proven for: origins up to ~6 W from the base
                std::vector<float> far(N * 6), tf(N); std::vector<uint32_t> pf(N);
                float ext = 0.f;
                for (int a = 0; a < 3; ++a) ext = std::fmax(ext, h.bounds_hi[a] - h.bounds_lo[a]);
                for (uint64_t i = 0; i < N; ++i) {
                    const uint64_t tri = (uint64_t)(rnd() * 0.999f * T);
                    float c[3], u[3], len = 0.f;
                    for (int a = 0; a < 3; ++a) { c[a] = (v[tri * 9 + a] + v[tri * 9 + 3 + a] + v[tri * 9 + 6 + a]) / 3.f; u[a] = rnd() - 0.5f; len += u[a] * u[a]; }
                    len = std::sqrt(len) + 1e-9f;
                    const float dist = rnd() * 6.f * ext;
                    for (int a = 0; a < 3; ++a) { far[i * 6 + 3 + a] = u[a] / len; far[i * 6 + a] = c[a] - u[a] / len * dist; }
                }
                orc_cast_brute(v.data(), f.data(), T, far.data(), N, tf.data(), pf.data());
                if (int rc = check_qnodes(h, far.data(), N, tf.data(), pf.data(), round)) return rc;
            }
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

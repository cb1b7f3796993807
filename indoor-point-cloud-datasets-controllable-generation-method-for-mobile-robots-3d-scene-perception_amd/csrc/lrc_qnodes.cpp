// lrc_qnodes.cpp -- see lrc_qnodes.h
#include "lrc_qnodes.h"

#include <cmath>
#include <cstring>

namespace lrc {

// ---- quantised node images ----------------------------------------------------------------
// Per axis a power-of-two width W and a float32 base b put the scene into normalised coordinates n = (x - b) / W in
// [2, 4): every float32 there has the exponent byte 0x40, so a 15-bit grid index q IS the float 0x40000000 | q << 8
// = 2 + q * 2^-14 and one v_perm_b32 rebuilds it from a packed half word.  A child box is rounded outward to the grid
// with a margin of 1/16 cell: the allowance for the difference between the slab arithmetic in normalised coordinates
// and the definition's arithmetic in world coordinates, which is below 2^-24 (4|o - b| + |o| + 8 W) in box space, i.e.
// below 2.7e-6 W = 0.044 cell for a ray origin o with |o - b| <= 6 W and |o| <= 12 W (the scene itself is [b + 2 W,
// b + 4 W), so this holds one scene width around the scene); the kernel sends every other ray through the float32
// world-space nodes.  Returns false (no images; the float32 nodes are used) when a scene does not fit.
bool make_qgrid(const HostBVH& h, float base[3], float W[3], float invW[3], QGrid& g) {
    if (h.num_nodes == 0) return false;
    return make_qgrid_bounds(h.bounds_lo, h.bounds_hi, base, W, invW, g);
}

bool make_qgrid_bounds(const float blo[3], const float bhi[3], float base[3], float W[3], float invW[3], QGrid& g) {
    for (int a = 0; a < 3; ++a) {
        const double lo = blo[a], hi = bhi[a];
        if (!(hi >= lo)) return false;
        int k = -20;                                                 // W = 2^k, 2^-20 <= W <= 2^16
        while (k <= 16 && 2.0 * std::ldexp(1.0, k) * (1.0 - 1.0 / 1024) < (hi - lo)) ++k;
        if (k > 16) return false;
        g.Wd[a] = std::ldexp(1.0, k);
        const double cell = g.Wd[a] / 16384.0;
        const double b = lo - 8.0 * cell - 2.0 * g.Wd[a];
        float bf = (float)b;
        if ((double)bf > b) bf = std::nextafter(bf, -INFINITY);
        g.bd[a] = (double)bf;
        if (!(std::fabs(g.bd[a]) <= kQnodeMaxBase * g.Wd[a])) return false;    // a scene this far from the world origin: all rays "far"
        base[a] = bf; W[a] = (float)g.Wd[a]; invW[a] = (float)(1.0 / g.Wd[a]);
    }
    return true;
}

// grid indices of a float32 box, rounded outward with the margin; false: outside the grid
static bool qbox(const QGrid& g, const float* lo, const float* hi, uint32_t ql[3], uint32_t qh[3]) {
    for (int a = 0; a < 3; ++a) {
        const double nl = ((double)lo[a] - g.bd[a]) / g.Wd[a], nh = ((double)hi[a] - g.bd[a]) / g.Wd[a];   // exact
        const double fl = std::floor((nl - 2.0) * 16384.0 - kQnodeMargin), fh = std::ceil((nh - 2.0) * 16384.0 + kQnodeMargin);
        if (!(fl >= 0.0) || !(fh <= 32767.0) || !(fl <= fh)) return false;
        ql[a] = (uint32_t)fl; qh[a] = (uint32_t)fh;
    }
    return true;
}

static float qdecode(uint32_t q) { uint32_t u = 0x40000000u | (q << 8); float f; std::memcpy(&f, &u, 4); return f; }

bool build_qnodes(const HostBVH& h, const QGrid& g, std::vector<uint32_t>& q8, std::vector<float>& n16,
                         double* leaf_inflation) {
    const uint64_t N = h.num_nodes;
    q8.assign(N * 8, 0u);
    n16.assign(N * 16, 0.0f);
    double infl_sum = 0.0;       // over leaf boxes: half perimeter of the quantised box / of the float32 box
    uint64_t infl_n = 0;
    for (uint64_t i = 0; i < N; ++i) {
        const float* nd = h.nodes.data() + i * 16;
        uint32_t* qo = q8.data() + i * 8;
        float* no = n16.data() + i * 16;
        for (int c = 0; c < 2; ++c) {
            const float* lo = nd + c * 6, *hi = nd + c * 6 + 3;
            uint32_t ql[3] = {32767u, 32767u, 32767u}, qh[3] = {0u, 0u, 0u};      // the empty leaf: inverted, never hit
            int32_t cref;
            std::memcpy(&cref, &nd[12 + c], 4);
            if (cref != ~0 && !qbox(g, lo, hi, ql, qh)) return false;
            for (int a = 0; a < 3; ++a) {
                qo[c * 4 + a] = ql[a] | (qh[a] << 16);
                no[c * 6 + a] = qdecode(ql[a]);
                no[c * 6 + 3 + a] = qdecode(qh[a]);
            }
            std::memcpy(&qo[c * 4 + 3], &nd[12 + c], 4);             // the child reference
            int32_t ref;
            std::memcpy(&ref, &nd[12 + c], 4);
            if (ref < 0 && ref != ~0) {
                double hw = 0.0, hq = 0.0;
                for (int a = 0; a < 3; ++a) {
                    hw += (double)hi[a] - (double)lo[a];
                    hq += (double)(qh[a] - ql[a]) * (g.Wd[a] / 16384.0);
                }
                if (hw > 0.0) { infl_sum += hq / hw; infl_n += 1; }
            }
        }
        no[12] = nd[12]; no[13] = nd[13];
    }
    *leaf_inflation = infl_n ? infl_sum / (double)infl_n : 1.0;
    return true;
}

// The binary tree collapsed to four children per node: a node keeps the children of its inner children (its grandchildren)
// and its leaf children; every second level disappears.  Slots [0, 1] come from child 0, [2, 3] from child 1; an unused
// slot holds an inverted box (never hit) and the empty leaf.  References of inner slots index THIS array (breadth-first
// numbering), leaf references are unchanged.  Same grid, same outward rounding as build_qnodes.
bool build_q4nodes(const HostBVH& h, const QGrid& g, std::vector<uint32_t>& q16, std::vector<float>& n32,
                          uint64_t* num4) {
    const uint64_t N = h.num_nodes;
    std::vector<uint32_t> order;              // binary node of every 4-wide node, in 4-wide numbering
    order.reserve(N / 2 + 1);
    order.push_back(0u);
    q16.clear(); n32.clear();
    q16.reserve((N / 2 + 1) * 16); n32.reserve((N / 2 + 1) * 32);
    const int32_t empty_ref = ~0;
    for (size_t i4 = 0; i4 < order.size(); ++i4) {
        const float* X = h.nodes.data() + (size_t)order[i4] * 16;
        struct Slot { const float* lo; const float* hi; int32_t ref; bool inner; } slot[4];
        for (int c = 0; c < 2; ++c) {
            int32_t ref;
            std::memcpy(&ref, &X[12 + c], 4);
            Slot& s0 = slot[2 * c]; Slot& s1 = slot[2 * c + 1];
            if (ref >= 0) {                   // inner child: its two children take the two slots
                const float* Y = h.nodes.data() + (size_t)ref * 16;
                for (int cc = 0; cc < 2; ++cc) {
                    Slot& t = cc ? s1 : s0;
                    t.lo = Y + cc * 6; t.hi = Y + cc * 6 + 3;
                    std::memcpy(&t.ref, &Y[12 + cc], 4);
                    t.inner = t.ref >= 0;
                }
            } else {                          // leaf child (or the empty leaf): stays, the other slot is unused
                s0.lo = ref == empty_ref ? nullptr : X + c * 6; s0.hi = X + c * 6 + 3; s0.ref = ref; s0.inner = false;
                s1.lo = nullptr; s1.hi = nullptr; s1.ref = empty_ref; s1.inner = false;
            }
        }
        const size_t qo = q16.size(), no = n32.size();
        q16.resize(qo + 16, 0u);
        n32.resize(no + 32, 0.0f);
        for (int k = 0; k < 4; ++k) {
            uint32_t ql[3] = {32767u, 32767u, 32767u}, qh[3] = {0u, 0u, 0u};          // inverted: never hit
            int32_t ref = slot[k].ref;
            if (slot[k].lo) {
                if (!qbox(g, slot[k].lo, slot[k].hi, ql, qh)) return false;
                if (slot[k].inner) {
                    if (order.size() >= 0x7FFFFFFFull) return false;
                    const int32_t child4 = (int32_t)order.size();
                    order.push_back((uint32_t)ref);
                    ref = child4;
                }
            }
            for (int a = 0; a < 3; ++a) {
                q16[qo + k * 4 + a] = ql[a] | (qh[a] << 16);
                n32[no + k * 8 + a] = qdecode(ql[a]);
                n32[no + k * 8 + 3 + a] = qdecode(qh[a]);
            }
            std::memcpy(&q16[qo + k * 4 + 3], &ref, 4);
            std::memcpy(&n32[no + k * 8 + 6], &ref, 4);
        }
    }
    *num4 = order.size();
    return true;
}

}  // namespace lrc

// bvh_build.cpp -- binned-SAH binary BVH over a triangle mesh, built once per mesh on the host.
//
// Product code of liblidarcast.  Replaces the Embree build hidden in
// open3d RaycastingScene.add_triangles (reference: raycast_engine/raycast_engine_cpu.py:46-47),
// which the reference repeats for every pose (raycast_engine/raycast_engine.py:20-24).
//
// Properties the traversal kernels rely on (tests/test_parity_gpu.py::test_bvh_invariants checks them on exported trees):
//   * every triangle sits in exactly one leaf slot; leaves hold 1..max_leaf triangles;
//   * a child box is the EXACT float32 min/max of the vertices below it (no arithmetic, no padding),
//     so computed slab intervals nest ("why any BVH gives the same hits");
//   * leaf depth <= 31, so a 32-entry traversal stack can never overflow;
//   * the first `bfs_nodes` nodes are in breadth-first order (top of tree contiguous: the part every wave walks, fetched through the scalar cache),
//     the rest in depth-first order (a subtree is contiguous in HBM);
//   * deterministic for a given input.
#include "lrc_bvh.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <deque>
#include <future>
#include <thread>

namespace lrc {
namespace {

struct Prim {
    float lo[3], hi[3], c[3];
    uint32_t id;
};

// min / max under the total order of the float32 bit patterns (-0.0 < +0.0): a box plane does not depend on the
// order its members were visited in, so the device builder (lrc_bvh_device.hip, integer min / max on the same
// encoding) produces the same bytes.  The sign of a zero plane never changes a traversal result.
inline int32_t ford(float f) { int32_t i; std::memcpy(&i, &f, 4); return i ^ ((i >> 31) & 0x7FFFFFFF); }
inline float fmin_t(float a, float b) { return ford(b) < ford(a) ? b : a; }
inline float fmax_t(float a, float b) { return ford(b) > ford(a) ? b : a; }

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; }
    }
    void grow(const float* l, const float* h) {
        for (int k = 0; k < 3; ++k) { lo[k] = fmin_t(lo[k], l[k]); hi[k] = fmax_t(hi[k], h[k]); }
    }
    void grow_pt(const float* p) {
        for (int k = 0; k < 3; ++k) { lo[k] = fmin_t(lo[k], p[k]); hi[k] = fmax_t(hi[k], p[k]); }
    }
    double half_area() const {
        double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct TmpNode {
    Box box[2];
    int64_t child[2];   // >= 0: tmp inner index; < 0: leaf, ~child = first_slot*8 + count
};

#ifndef LRC_BINS
#define LRC_BINS 64
#endif
constexpr int kBins = LRC_BINS;

inline int ceil_log2_u64(uint64_t x) {   // smallest k with 2^k >= x  (x >= 1)
    int k = 0;
    uint64_t p = 1;
    while (p < x) { p <<= 1; ++k; }
    return k;
}

// The recursion forks std::async tasks near the root (disjoint prim ranges, nodes taken from one
// pre-sized pool through an atomic counter).  Temporary node numbers then depend on thread timing, but
// the TREE does not, and the final relayout renumbers nodes from the structure alone: the output is
// deterministic.
struct Builder {
    std::vector<Prim> prims;
    std::vector<TmpNode> nodes;              // pool, sized up front
    std::atomic<int64_t> next_node{0};
    int max_leaf = kMaxLeaf;
    bool median_only = false;
    int fork_depth = 0;                      // fork while depth < fork_depth
    int depth_cap = kMaxDepth - 1;           // deepest allowed leaf (<= kMaxDepth - 1)
    std::atomic<uint32_t> max_depth{0};
    std::atomic<uint32_t> max_leaf_seen{0};
    std::atomic<uint64_t> num_leaves{0};

    static void atomic_max(std::atomic<uint32_t>& a, uint32_t v) {
        uint32_t cur = a.load(std::memory_order_relaxed);
        while (cur < v && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
    }

    // height of a median-split subtree over n prims (0 = it is a single leaf)
    int median_height(uint64_t n) const {
        uint64_t leaves = (n + max_leaf - 1) / max_leaf;
        return ceil_log2_u64(leaves ? leaves : 1);
    }
    // can a subtree with n prims whose root sits at `depth` finish with leaf depth <= kMaxDepth-1 ?
    bool fits(uint64_t n, int depth) const { return depth + median_height(n) <= depth_cap; }

    int64_t make_leaf(uint64_t begin, uint64_t end, int depth) {
        uint64_t cnt = end - begin;
        // slots of a leaf in triangle-row order: the layout is then a function of the tree alone
        std::sort(prims.begin() + begin, prims.begin() + end, [](const Prim& a, const Prim& b) { return a.id < b.id; });
        atomic_max(max_depth, (uint32_t)depth);
        atomic_max(max_leaf_seen, (uint32_t)cnt);
        num_leaves.fetch_add(1, std::memory_order_relaxed);
        return ~(int64_t)(begin * 8 + cnt);
    }

    Box range_box(uint64_t begin, uint64_t end) const {
        Box b; b.reset();
        for (uint64_t i = begin; i < end; ++i) b.grow(prims[i].lo, prims[i].hi);
        return b;
    }

    // returns the child reference for prims[begin,end) whose root sits at `depth`
    int64_t build(uint64_t begin, uint64_t end, int depth) {
        const uint64_t n = end - begin;
        if (n <= (uint64_t)max_leaf) return make_leaf(begin, end, depth);

        Box cb; cb.reset();
        for (uint64_t i = begin; i < end; ++i) cb.grow_pt(prims[i].c);

        uint64_t mid = 0;
        bool have_split = false;

        // ---- binned SAH over the three axes ----
        double best_cost = DBL_MAX;
        int best_axis = -1, best_bin = -1;
        for (int ax = 0; ax < 3 && !median_only; ++ax) {
            const float ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.0f)) continue;
            const float scale = (float)kBins / ext;
            Box bb[kBins];
            uint64_t cnt[kBins];
            for (int b = 0; b < kBins; ++b) { bb[b].reset(); cnt[b] = 0; }
            for (uint64_t i = begin; i < end; ++i) {
                int b = (int)((prims[i].c[ax] - cb.lo[ax]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bb[b].grow(prims[i].lo, prims[i].hi);
                ++cnt[b];
            }
            double right_area[kBins];
            uint64_t right_cnt[kBins];
            Box acc; acc.reset();
            uint64_t c = 0;
            for (int b = kBins - 1; b >= 1; --b) {
                if (cnt[b]) acc.grow(bb[b].lo, bb[b].hi);
                c += cnt[b];
                right_area[b] = acc.half_area();
                right_cnt[b] = c;
            }
            acc.reset(); c = 0;
            for (int b = 0; b < kBins - 1; ++b) {   // split between bin b and b+1
                if (cnt[b]) acc.grow(bb[b].lo, bb[b].hi);
                c += cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                double cost = acc.half_area() * (double)c + right_area[b + 1] * (double)right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
            }
        }
        if (best_axis >= 0) {
            const int ax = best_axis;
            const float scale = (float)kBins / (cb.hi[ax] - cb.lo[ax]);
            const float lo = cb.lo[ax];
            const int bin = best_bin;
            auto it = std::partition(prims.begin() + begin, prims.begin() + end, [&](const Prim& p) {
                int b = (int)((p.c[ax] - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= bin;
            });
            mid = (uint64_t)(it - prims.begin());
            have_split = mid > begin && mid < end &&
                         fits(mid - begin, depth + 1) && fits(end - mid, depth + 1);
        }
        if (!have_split) {
            // median split on the widest centroid axis (also the depth-guard fallback)
            int ax = 0;
            float e0 = cb.hi[0] - cb.lo[0], e1 = cb.hi[1] - cb.lo[1], e2 = cb.hi[2] - cb.lo[2];
            if (e1 > e0 && e1 >= e2) ax = 1; else if (e2 > e0 && e2 > e1) ax = 2;
            mid = begin + (n + 1) / 2;
            std::nth_element(prims.begin() + begin, prims.begin() + mid, prims.begin() + end,
                             [ax](const Prim& a, const Prim& b) {
                                 if (a.c[ax] != b.c[ax]) return a.c[ax] < b.c[ax];
                                 return a.id < b.id;
                             });
        }

        const int64_t me = next_node.fetch_add(1, std::memory_order_relaxed);
        nodes[me].box[0] = range_box(begin, mid);
        nodes[me].box[1] = range_box(mid, end);
        int64_t c0, c1;
        if (depth < fork_depth && n >= 20000) {
            auto left = std::async(std::launch::async, [this, begin, mid, depth] { return build(begin, mid, depth + 1); });
            c1 = build(mid, end, depth + 1);
            c0 = left.get();
        } else {
            c0 = build(begin, mid, depth + 1);
            c1 = build(mid, end, depth + 1);
        }
        nodes[me].child[0] = c0;
        nodes[me].child[1] = c1;
        return me;
    }
};

inline float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

}  // namespace

void build_bvh(const float* verts3, uint64_t V, const uint32_t* tris3, uint64_t T,
               const uint16_t* tri_sem, const uint16_t* tri_ins,
               const BuildOptions& opt, HostBVH* out) {
    (void)V;
    HostBVH& o = *out;
    o = HostBVH();
    if (T == 0) return;

    Builder b;
    b.max_leaf = std::min(std::max(opt.max_leaf, 1), kMaxLeaf);
    b.median_only = opt.median_only != 0;
    b.prims.resize(T);
    Box all; all.reset();
    for (uint64_t t = 0; t < T; ++t) {
        Prim& p = b.prims[t];
        p.id = (uint32_t)t;
        Box bx; bx.reset();
        for (int j = 0; j < 3; ++j) bx.grow_pt(verts3 + 3 * (uint64_t)tris3[3 * t + j]);
        for (int k = 0; k < 3; ++k) {
            p.lo[k] = bx.lo[k]; p.hi[k] = bx.hi[k];
            p.c[k] = 0.5f * bx.lo[k] + 0.5f * bx.hi[k];
        }
        all.grow(bx.lo, bx.hi);
    }
    for (int k = 0; k < 3; ++k) { o.bounds_lo[k] = all.lo[k]; o.bounds_hi[k] = all.hi[k]; }

    b.nodes.resize(T + 1);                   // inner nodes < leaves <= T
    {
        // every level of leaf depth costs 256 B of LDS stack per wave in the trace kernel; SAH trees of gridded
        // meshes are only a few levels deeper than the balanced tree, so cap the depth at balanced + slack
        const int balanced = b.median_height(T);
        int cap = opt.depth_slack >= 0 ? balanced + opt.depth_slack : kMaxDepth - 1;
        b.depth_cap = std::min(std::max(cap, balanced), kMaxDepth - 1);
    }
    {
        unsigned hw = std::thread::hardware_concurrency();
        int threads = opt.threads > 0 ? opt.threads : (int)std::min<unsigned>(hw ? hw : 1u, 16u);
        while ((1 << b.fork_depth) < threads) ++b.fork_depth;      // 2^fork_depth tasks
    }
    int64_t root = b.build(0, T, 0);
    b.nodes.resize((size_t)b.next_node.load());
    if (root < 0) {
        // the whole mesh is one leaf: wrap it so that node 0 is always an inner node
        TmpNode nd;
        nd.box[0] = all;
        std::memset(&nd.box[1], 0, sizeof(Box));
        nd.child[0] = root;
        nd.child[1] = ~(int64_t)0;   // empty leaf: slot 0, count 0
        b.nodes.push_back(nd);
        Builder::atomic_max(b.max_depth, 1);
        root = 0;
    }

    // ---- relayout: breadth-first head, depth-first tail ----
    const uint64_t nn = b.nodes.size();
    std::vector<int64_t> new_of_old(nn, -1);
    std::vector<int64_t> old_of_new;
    old_of_new.reserve(nn);
    std::deque<int64_t> q;
    q.push_back(root);
    const uint64_t head = (uint64_t)std::max(opt.bfs_nodes, 1);
    while (!q.empty() && old_of_new.size() < head) {
        int64_t n = q.front(); q.pop_front();
        new_of_old[n] = (int64_t)old_of_new.size();
        old_of_new.push_back(n);
        for (int c = 0; c < 2; ++c) if (b.nodes[n].child[c] >= 0) q.push_back(b.nodes[n].child[c]);
    }
    std::vector<int64_t> st;
    while (!q.empty()) {
        st.push_back(q.front()); q.pop_front();
        while (!st.empty()) {
            int64_t n = st.back(); st.pop_back();
            new_of_old[n] = (int64_t)old_of_new.size();
            old_of_new.push_back(n);
            if (b.nodes[n].child[1] >= 0) st.push_back(b.nodes[n].child[1]);
            if (b.nodes[n].child[0] >= 0) st.push_back(b.nodes[n].child[0]);
        }
    }

    o.num_nodes = nn;
    o.num_leaves = b.num_leaves.load();
    o.num_slots = T;
    o.max_depth = b.max_depth.load();
    o.max_leaf_size = b.max_leaf_seen.load();
    o.nodes.assign(nn * kNodeFloats, 0.0f);
    for (uint64_t i = 0; i < nn; ++i) {
        const TmpNode& nd = b.nodes[old_of_new[i]];
        float* f = &o.nodes[i * kNodeFloats];
        for (int c = 0; c < 2; ++c) {
            for (int k = 0; k < 3; ++k) { f[c * 6 + k] = nd.box[c].lo[k]; f[c * 6 + 3 + k] = nd.box[c].hi[k]; }
            int32_t ref = nd.child[c] >= 0 ? (int32_t)new_of_old[nd.child[c]] : (int32_t)nd.child[c];
            std::memcpy(&f[12 + c], &ref, 4);
        }
    }

    // ---- triangle records in slot (leaf) order ----
    o.tri_rec.assign((T + kMaxLeaf - 1) * kTriFloats, 0.0f);   // padded: leaf-wide reads may overrun a short last leaf
    o.slot_prim.resize(T);
    o.slot_label.resize(T);
    for (uint64_t s = 0; s < T; ++s) {
        const uint32_t id = b.prims[s].id;
        const float* v0 = verts3 + 3 * (uint64_t)tris3[3 * (uint64_t)id + 0];
        const float* v1 = verts3 + 3 * (uint64_t)tris3[3 * (uint64_t)id + 1];
        const float* v2 = verts3 + 3 * (uint64_t)tris3[3 * (uint64_t)id + 2];
        float e1[3], e2[3];
        for (int k = 0; k < 3; ++k) { e1[k] = v0[k] - v1[k]; e2[k] = v2[k] - v0[k]; }
        float* r = &o.tri_rec[s * kTriFloats];
        for (int k = 0; k < 3; ++k) { r[k] = v0[k]; r[3 + k] = e1[k]; r[6 + k] = e2[k]; }
        if (o.slot_box.empty()) o.slot_box.assign(T * 6, 0.0f);
        for (int k = 0; k < 3; ++k) {
            o.slot_box[s * 6 + k] = fmin_t(fmin_t(v0[k], v1[k]), v2[k]);
            o.slot_box[s * 6 + 3 + k] = fmax_t(fmax_t(v0[k], v1[k]), v2[k]);
        }
        // Ng = cross(e2, e1), component = fma(a_j, b_k, -(a_k * b_j))
        r[9]  = fmaf_(e2[1], e1[2], -(e2[2] * e1[1]));
        r[10] = fmaf_(e2[2], e1[0], -(e2[0] * e1[2]));
        r[11] = fmaf_(e2[0], e1[1], -(e2[1] * e1[0]));
        o.slot_prim[s] = id;
        uint32_t sem = tri_sem ? tri_sem[id] : 0u, ins = tri_ins ? tri_ins[id] : 0u;
        o.slot_label[s] = sem | (ins << 16);
    }
}

}  // namespace lrc

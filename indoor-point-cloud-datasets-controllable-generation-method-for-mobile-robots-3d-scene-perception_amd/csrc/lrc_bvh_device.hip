// lrc_bvh_device.hip -- the scene build on the GPU (gfx950 only); see lrc_bvh_device.h.
//
// Replaces, once per mesh, the Embree build hidden in open3d RaycastingScene.add_triangles (reference call site
// raycast_engine/raycast_engine_cpu.py:46-47, repeated per pose by raycast_engine/raycast_engine.py:20-24).
//
// The algorithm is bvh_build.cpp's, restated level by level:
//   * a level = all inner nodes of one depth, in left-to-right order; every node owns a contiguous range of the
//     primitive array (32-byte records: box + triangle row), which its split partitions into the other buffer;
//   * nodes of <= 16 primitives -- most nodes of the last levels -- four to a wave (k_tiny: a candidate split per
//     primitive instead of per bin: the split "after bin b" changes only at occupied bins, so the occupied bins are the
//     candidates; same costs, same first minimum), nodes of <= 64 primitives by ONE WAVE each (k_small: primitive per lane, the 64 bins of an axis in
//     LDS, then bin per lane: prefix / suffix of boxes and counts by wave shuffles, costs in double precision,
//     arg-min by butterfly), nodes of <= 1024 by one workgroup (k_medium: three waves evaluate the three axes side by
//     side), larger ones by workgroups per 1024-primitive chunk with the bins of a node combined by integer atomics
//     (k_big_bin / k_big_eval / k_big_scatter);
//   * the median fallback (no SAH split, or a split the depth cap forbids) ranks the primitives by (centroid, row): in
//     registers / LDS for small and medium nodes, by a radix sort of the segment for a big one;
//   * k_emit numbers the children (an ordered scan: breadth-first numbering), k_write_nodes lays the tree out as the
//     host does (breadth-first head, depth-first tail: the rank of (begin, depth) among the tail nodes) and rounds the
//     quantised node images; k_tri_records writes the 48-byte triangle records, ids, labels and the plane table.
// Every quantity a decision depends on is order independent: min / max under the total order of the float bit
// patterns, integer counts, and double-precision costs formed per bin exactly as the host's sweep forms them.  Hence
// the same tree as the host builder, and -- both builders sort a leaf's slots by triangle row -- the same bytes.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "../../include/lidarcast.h"
#include "lrc_bvh_device.h"
#include "lrc_qnodes.h"

namespace lrc {
namespace {

constexpr int kTinyMax = 16;         // four nodes per wave (16 lanes each)
constexpr int kSmallMax = 64;        // one wave per node
constexpr int kMediumMax = 1024;     // one workgroup per node
constexpr int kChunk = 1024;         // primitives per workgroup of the big-node kernels
constexpr int kBinsN = 64;           // == LRC_BINS of bvh_build.cpp
constexpr int kMaxLevels = 64;
constexpr int kEncPosMax = 0x7F7FFFFF;           // enc(+FLT_MAX)
constexpr int kEncNegMax = (int)0x80800000u;     // enc(-FLT_MAX)

#define DI __device__ __forceinline__

// float <-> int with the same order (and -0.0 < +0.0): min / max become integer min / max, associative and
// commutative, so a box does not depend on the order its members arrive in
DI int enc(float f) { const int i = __float_as_int(f); return i ^ ((i >> 31) & 0x7FFFFFFF); }
DI float dec(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7FFFFFFF)); }
DI int imin(int a, int b) { return a < b ? a : b; }
DI int imax(int a, int b) { return a > b ? a : b; }

struct PrimRec { float4 a, b; };     // a = lo.xyz, hi.x   b = hi.y, hi.z, triangle row (bits), 0

struct TNode {                       // 128 B, by breadth-first index
    uint32_t begin, end, mid, depth;
    int32_t child[2];
    int32_t work;                    // big nodes: slot of their BigWork record
    uint32_t cb_valid;
    int32_t cb[6];                   // centroid bounds (ordered ints): lo xyz, hi xyz
    int32_t box[2][6];               // child boxes (ordered ints)
    uint32_t pad[6];
};
static_assert(sizeof(TNode) == 128, "TNode layout");

struct BigWork {                     // one per big node of the current level
    uint32_t g;
    int32_t axis, bin;               // SAH split: primitives with bin(c[axis]) <= bin go left
    float lo, scale;
    uint32_t median;                 // 1: median split along `axis` instead
    uint32_t cur_left, cur_right;    // allocation cursors of the scatter
    int32_t ccb[2][6];               // centroid bounds of the two children
    uint32_t pad[4];
};

struct LevelCounters { uint32_t n_nodes, n_small, n_medium, n_big, n_leaves, max_leaf, n_median, error, n_tiny, n_sub, pad[2]; };

struct Params {
    const PrimRec* cur;
    PrimRec* nxt;
    TNode* nodes;
    uint32_t* final_id;
    int max_leaf, depth_cap, median_only;
};

DI int median_height(uint32_t n, int max_leaf) {
    const uint32_t leaves = (n + (uint32_t)max_leaf - 1u) / (uint32_t)max_leaf;
    return leaves <= 1u ? 0 : 32 - __builtin_clz(leaves - 1u);
}
DI bool fits(uint32_t n, int depth, const Params& P) { return depth + median_height(n, P.max_leaf) <= P.depth_cap; }

DI int wave_min(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = imin(v, __shfl_xor(v, d, 64));
    return v;
}
DI int wave_max(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = imax(v, __shfl_xor(v, d, 64));
    return v;
}

DI double half_area(const int b[6]) {
    const double dx = (double)dec(b[3]) - (double)dec(b[0]), dy = (double)dec(b[4]) - (double)dec(b[1]),
                 dz = (double)dec(b[5]) - (double)dec(b[2]);
    if (dx < 0 || dy < 0 || dz < 0) return 0.0;
    return dx * dy + dy * dz + dz * dx;
}

struct AxisBest {
    double cost;          // +inf: this axis has no valid split
    int bin;
    int lbox[6], rbox[6];
    uint32_t lcnt, rcnt;
};

// One axis, evaluated by one wave: lane b holds bin b (box as ordered ints, count).  Reproduces the two sweeps of
// bvh_build.cpp: split "after bin b" has the union of bins 0..b on the left and b+1..63 on the right, cost =
// area(L) * count(L) + area(R) * count(R) in double; the first minimum in bin order wins.  Every lane returns the result.
DI AxisBest eval_axis(int lane, const int bin_box[6], uint32_t bin_cnt) {
    int L[6], R[6];
    uint32_t LC = bin_cnt, RC = bin_cnt;
#pragma unroll
    for (int k = 0; k < 6; ++k) { L[k] = bin_box[k]; R[k] = bin_box[k]; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int v = __shfl_up(L[k], d, 64), w = __shfl_down(R[k], d, 64);
            if (lane >= d) L[k] = k < 3 ? imin(L[k], v) : imax(L[k], v);
            if (lane + d < 64) R[k] = k < 3 ? imin(R[k], w) : imax(R[k], w);
        }
        const uint32_t vc = __shfl_up(LC, d, 64), wc = __shfl_down(RC, d, 64);
        if (lane >= d) LC += vc;
        if (lane + d < 64) RC += wc;
    }
    int Rn[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Rn[k] = __shfl_down(R[k], 1, 64);
    const uint32_t RCn = __shfl_down(RC, 1, 64);
    const bool valid = (lane < 63) & (LC > 0u) & (RCn > 0u);
    double c = __builtin_inf();
    if (valid) c = half_area(L) * (double)LC + half_area(Rn) * (double)RCn;
    int bi = lane;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double oc = __shfl_xor(c, d, 64);
        const int ob = __shfl_xor(bi, d, 64);
        if (oc < c || (oc == c && ob < bi)) { c = oc; bi = ob; }
    }
    AxisBest r;
    r.cost = c;
    r.bin = bi;
#pragma unroll
    for (int k = 0; k < 6; ++k) { r.lbox[k] = __shfl(L[k], bi, 64); r.rbox[k] = __shfl(Rn[k], bi, 64); }
    r.lcnt = __shfl(LC, bi, 64);
    r.rcnt = __shfl(RCn, bi, 64);
    return r;
}

DI int bin_of(float c, float lo, float scale) {
    int b = (int)((c - lo) * scale);
    return b < 0 ? 0 : (b >= kBinsN ? kBinsN - 1 : b);
}

// median fallback: the split axis, by bvh_build.cpp's rule on the centroid extents
DI int median_axis(const int cb[6]) {
    const float e0 = dec(cb[3]) - dec(cb[0]), e1 = dec(cb[4]) - dec(cb[1]), e2 = dec(cb[5]) - dec(cb[2]);
    int ax = 0;
    if (e1 > e0 && e1 >= e2) ax = 1; else if (e2 > e0 && e2 > e1) ax = 2;
    return ax;
}
DI bool key_less(float ca, uint32_t ia, float cb, uint32_t ib) { return ca != cb ? ca < cb : ia < ib; }

struct LoadedPrim { float lo[3], hi[3], c[3]; uint32_t id; };
DI LoadedPrim load_prim(const PrimRec* p) {
    const float4 a = p->a, b = p->b;
    LoadedPrim r;
    r.lo[0] = a.x; r.lo[1] = a.y; r.lo[2] = a.z; r.hi[0] = a.w; r.hi[1] = b.x; r.hi[2] = b.y;
    r.id = __float_as_uint(b.z);
#pragma unroll
    for (int k = 0; k < 3; ++k) r.c[k] = 0.5f * r.lo[k] + 0.5f * r.hi[k];
    return r;
}
DI void store_prim(PrimRec* p, const LoadedPrim& r) {
    p->a = make_float4(r.lo[0], r.lo[1], r.lo[2], r.hi[0]);
    p->b = make_float4(r.hi[1], r.hi[2], __uint_as_float(r.id), 0.0f);
}

// ---- setup: primitive records, scene bounds, root centroid bounds, input validation --------------------------------
// The state a build starts from, in one launch (it used to be three memsets, seven four-byte copies and a synchronisation):
// all level counters zero, level 0 = the root alone in the list of its size class, empty scene bounds, the root node.
struct SetupArgs { TNode* nodes; LevelCounters* counters; int* bounds; uint32_t* qfail; uint32_t* sub_count; uint32_t* list0[4]; uint32_t T; };
__global__ __launch_bounds__(256) void k_setup(const SetupArgs a) {
    const uint32_t tid = threadIdx.x;
    uint32_t* cw = (uint32_t*)a.counters;
    for (uint32_t i = tid; i < (uint32_t)(sizeof(LevelCounters) / 4 * (kMaxLevels + 1)); i += 256) cw[i] = 0u;
    if (tid < 8) { a.bounds[tid] = tid < 3 ? kEncPosMax : (tid < 6 ? kEncNegMax : 0); a.sub_count[tid] = 0u; }
    if (tid < 4) { a.qfail[tid] = 0u; a.list0[tid][0] = 0u; }
    __syncthreads();
    if (tid != 0) return;
    TNode root{};
    root.begin = 0; root.end = a.T; root.mid = 0; root.depth = 0; root.work = -1; root.cb_valid = 1;
    for (int k = 0; k < 6; ++k) root.cb[k] = k < 3 ? kEncPosMax : kEncNegMax;
    a.nodes[0] = root;
    LevelCounters& l0 = a.counters[0];
    l0.n_nodes = 1;
    if (a.T <= (uint32_t)kTinyMax) l0.n_tiny = 1; else if (a.T <= (uint32_t)kSmallMax) l0.n_small = 1;
    else if (a.T <= (uint32_t)kMediumMax) l0.n_medium = 1; else l0.n_big = 1;
}

__global__ __launch_bounds__(256) void k_check_verts(const float* verts3, uint64_t n3, LevelCounters* lc) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    if ((__float_as_uint(verts3[i]) & 0x7FFFFFFFu) > 0x49742400u /* 1e6f */) atomicOr(&lc->error, 2u);
}

__global__ __launch_bounds__(256) void k_init_prims(const float* verts3, uint64_t V, const uint32_t* tris3, uint32_t T,
                                                    PrimRec* prims, TNode* root, int* bounds, LevelCounters* lc) {
    __shared__ int s_red[12];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 12) s_red[threadIdx.x] = (threadIdx.x % 6) < 3 ? kEncPosMax : kEncNegMax;
    __syncthreads();
    int red[12] = {kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax,
                   kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax};
    if (t < T) {
        const uint32_t i0 = tris3[3 * (size_t)t], i1 = tris3[3 * (size_t)t + 1], i2 = tris3[3 * (size_t)t + 2];
        if (i0 >= V || i1 >= V || i2 >= V) {
            atomicOr(&lc->error, 1u);
        } else {
            LoadedPrim r;
            r.id = t;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int a = enc(verts3[3 * (size_t)i0 + k]), b = enc(verts3[3 * (size_t)i1 + k]),
                          c = enc(verts3[3 * (size_t)i2 + k]);
                const int lo = imin(imin(a, b), c), hi = imax(imax(a, b), c);
                r.lo[k] = dec(lo); r.hi[k] = dec(hi);
                r.c[k] = 0.5f * r.lo[k] + 0.5f * r.hi[k];
                red[k] = lo; red[3 + k] = hi;
                red[6 + k] = red[9 + k] = enc(r.c[k]);
            }
            store_prim(prims + t, r);
        }
    }
    // scene bounds and the root's centroid bounds: wave butterfly, then one LDS atomic per wave and value
#pragma unroll
    for (int k = 0; k < 12; ++k) red[k] = (k % 6) < 3 ? wave_min(red[k]) : wave_max(red[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) { if ((k % 6) < 3) atomicMin(&s_red[k], red[k]); else atomicMax(&s_red[k], red[k]); }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        if (threadIdx.x < 3) { atomicMin(&bounds[threadIdx.x], s_red[threadIdx.x]); atomicMin(&root->cb[threadIdx.x], s_red[6 + threadIdx.x]); }
        else { atomicMax(&bounds[threadIdx.x], s_red[threadIdx.x]); atomicMax(&root->cb[threadIdx.x], s_red[6 + threadIdx.x]); }
    }
}

// ---- nodes of <= 64 primitives: one wave each ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_small(const Params P, const uint32_t* list, uint32_t n_list) {
    __shared__ int s_bins[4][7 * kBinsN];
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_list) return;                       // whole waves leave; no workgroup barrier below
    int* bins = s_bins[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const uint32_t g = list[w];
    TNode& nd = P.nodes[g];
    const uint32_t begin = nd.begin, n = nd.end - begin;
    const int depth = (int)nd.depth;
    const bool act = (uint32_t)lane < n;
    LoadedPrim pr{};
    if (act) pr = load_prim(P.cur + begin + lane);
    int cb[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cb[k] = wave_min(act ? enc(pr.c[k]) : kEncPosMax);
        cb[3 + k] = wave_max(act ? enc(pr.c[k]) : kEncNegMax);
    }
    double best_cost = 1.7976931348623157e308;     // DBL_MAX, as the host starts
    int best_axis = -1, best_bin = -1;
    int lbox[6], rbox[6];
    uint32_t nL = 0;
    int mybin[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k) { lbox[k] = 0; rbox[k] = 0; }
    if (!P.median_only) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float lo = dec(cb[ax]), ext = dec(cb[3 + ax]) - lo;
            if (!(ext > 0.0f)) continue;
            const float scale = (float)kBinsN / ext;
            const int b = bin_of(pr.c[ax], lo, scale);
            mybin[ax] = b;
#pragma unroll
            for (int k = 0; k < 3; ++k) { bins[k * kBinsN + lane] = kEncPosMax; bins[(3 + k) * kBinsN + lane] = kEncNegMax; }
            bins[6 * kBinsN + lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (act) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    atomicMin(&bins[k * kBinsN + b], enc(pr.lo[k]));
                    atomicMax(&bins[(3 + k) * kBinsN + b], enc(pr.hi[k]));
                }
                atomicAdd(&bins[6 * kBinsN + b], 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            int bb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) bb[k] = bins[k * kBinsN + lane];
            const uint32_t bc = (uint32_t)bins[6 * kBinsN + lane];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            const AxisBest r = eval_axis(lane, bb, bc);
            if (r.cost < best_cost) {
                best_cost = r.cost; best_axis = ax; best_bin = r.bin; nL = r.lcnt;
#pragma unroll
                for (int k = 0; k < 6; ++k) { lbox[k] = r.lbox[k]; rbox[k] = r.rbox[k]; }
            }
        }
    }
    bool pred = false;
    bool have_split = false;
    if (best_axis >= 0) {
        have_split = fits(nL, depth + 1, P) && fits(n - nL, depth + 1, P);
        const int mb = best_axis == 0 ? mybin[0] : (best_axis == 1 ? mybin[1] : mybin[2]);
        pred = mb <= best_bin;
    }
    if (!have_split) {
        const int ax = median_axis(cb);
        const float myc = ax == 0 ? pr.c[0] : (ax == 1 ? pr.c[1] : pr.c[2]);
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const float cj = __shfl(myc, (int)j, 64);
            const uint32_t ij = __shfl(pr.id, (int)j, 64);
            rank += key_less(cj, ij, myc, pr.id) ? 1u : 0u;
        }
        nL = (n + 1u) / 2u;
        pred = rank < nL;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lbox[k] = wave_min(act && pred ? enc(pr.lo[k]) : kEncPosMax);
            lbox[3 + k] = wave_max(act && pred ? enc(pr.hi[k]) : kEncNegMax);
            rbox[k] = wave_min(act && !pred ? enc(pr.lo[k]) : kEncPosMax);
            rbox[3 + k] = wave_max(act && !pred ? enc(pr.hi[k]) : kEncNegMax);
        }
    }
    const unsigned long long am = __ballot(act), lm = __ballot(act && pred);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (act) {
        const uint32_t pos = pred ? (uint32_t)__popcll(lm & below) : nL + (uint32_t)__popcll(am & ~lm & below);
        store_prim(P.nxt + begin + pos, pr);
        const uint32_t cn = pred ? nL : n - nL;
        if (cn <= (uint32_t)P.max_leaf) P.final_id[begin + pos] = pr.id;
    }
    if (lane == 0) {
        nd.mid = begin + nL;
#pragma unroll
        for (int k = 0; k < 6; ++k) { nd.box[0][k] = lbox[k]; nd.box[1][k] = rbox[k]; }
    }
}

// ---- whole subtrees of <= 64 primitives: one wave each (round 4) ------------------------------------------------------------
// Below a node of <= 64 primitives nothing needs the grid any more: the wave that holds its primitives (one per lane) splits
// it, keeps both halves in its lanes and goes on, node after node from a small stack in LDS, with k_small's arithmetic for
// every node (bins of the three axes in LDS, bin per lane, costs in double, first minimum; the median fallback by rank) --
// so the tree below such a node is the tree the level-by-level kernels built, and with it every byte of the scene.  What
// it saves is the grid: the last six to eight levels of a 10^6-triangle mesh (two to four launches and a counter read-back
// each) become ONE launch.  Descendants are staged at stage[first primitive + j] (an inner node per split position at most, so
// a root's nodes fit its own primitive range) with wave-local child numbers; k_sub_scan turns the per-root counts into node
// numbers behind the level-numbered nodes and k_sub_place moves them there.  Only nodes numbered beyond the breadth-first
// head of the final layout are built this way (the head keeps its breadth-first numbers; the tail is laid out by
// (first primitive, depth), whatever its numbers were).
struct SNode { uint32_t begin, end, mid, depth; int32_t child[2]; int32_t box[2][6]; uint32_t pad[2]; };
static_assert(sizeof(SNode) == 80, "SNode layout");
struct SubRoot { uint32_t g, parity; };              // node number of the subtree's root, buffer its primitives lie in
struct SubInfo { uint32_t count, leaves, max_leaf, max_depth; };     // per root: staged nodes, leaves, largest leaf, deepest inner node + 1
struct SubTotals { uint32_t nodes, leaves, max_leaf, max_depth; };

struct SubCand { double cost; int bin; int pad; };

// All nodes of one depth below the root are split AT ONCE: a lane holds the primitive at its position, knows the node
// [nb, ne) its position belongs to, and -- k_tiny's method, for segments of any length up to 64 -- evaluates ONE candidate,
// the split after the bin its own centroid falls into, by walking its node's primitives in LDS (lanes of one node read the
// same record: broadcast).  The host's sweep changes the partition only at occupied bins, equal partitions cost the same and
// the first minimum wins, so the minimum over a node's candidates with ties to the lower bin is the host's choice.
__global__ __launch_bounds__(256) void k_subtree(const Params P, const PrimRec* buf0, const PrimRec* buf1, const SubRoot* roots,
                                                 uint32_t n_roots, SNode* stage, SubInfo* info) {
    __shared__ uint4 s_rec[4][64 * 3];                // per position: (enc lo xyz, enc hi x) (enc hi y, enc hi z, bins, row) (c xyz, flag)
    __shared__ SubCand s_cand[4][64];
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_roots) return;                         // whole waves leave; no workgroup barrier below
    uint4* rec = s_rec[threadIdx.x >> 6];
    SubCand* cand = s_cand[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const SubRoot root = roots[w];
    TNode& rn = P.nodes[root.g];
    const uint32_t gbegin = rn.begin, n_root = rn.end - gbegin;
    LoadedPrim pr{};
    if ((uint32_t)lane < n_root) pr = load_prim((root.parity ? buf1 : buf0) + gbegin + lane);
    uint32_t nb = 0, ne = n_root, nid = 0;            // the node this POSITION belongs to, its wave-local number
    bool live = (uint32_t)lane < n_root;
    uint32_t next_local = 1, leaves = 0, maxl = 0;    // wave-uniform
    uint32_t depth = rn.depth;                        // all nodes split in one pass have the same depth
    const unsigned long long below = (1ull << lane) - 1ull;
    while (__builtin_amdgcn_ballot_w64(live)) {
        const uint32_t n = live ? ne - nb : 0u;
        rec[lane * 3] = make_uint4((uint32_t)enc(pr.lo[0]), (uint32_t)enc(pr.lo[1]), (uint32_t)enc(pr.lo[2]), (uint32_t)enc(pr.hi[0]));
        rec[lane * 3 + 2] = make_uint4(__float_as_uint(pr.c[0]), __float_as_uint(pr.c[1]), __float_as_uint(pr.c[2]), 0u);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        int cb[6] = {kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax};
        for (uint32_t t = 0; t < n; ++t) {
            const uint4 c = rec[(nb + t) * 3 + 2];
            const int e0 = enc(__uint_as_float(c.x)), e1 = enc(__uint_as_float(c.y)), e2 = enc(__uint_as_float(c.z));
            cb[0] = imin(cb[0], e0); cb[1] = imin(cb[1], e1); cb[2] = imin(cb[2], e2);
            cb[3] = imax(cb[3], e0); cb[4] = imax(cb[4], e1); cb[5] = imax(cb[5], e2);
        }
        int mybin[3] = {0, 0, 0};
        bool ok3[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float lo = dec(cb[ax]), ext = dec(cb[3 + ax]) - lo;
            ok3[ax] = live && (ext > 0.0f) && !P.median_only;
            mybin[ax] = ok3[ax] ? bin_of(pr.c[ax], lo, (float)kBinsN / ext) : 0;
        }
        rec[lane * 3 + 1] = make_uint4((uint32_t)enc(pr.hi[1]), (uint32_t)enc(pr.hi[2]),
                                       (uint32_t)mybin[0] | ((uint32_t)mybin[1] << 8) | ((uint32_t)mybin[2] << 16), pr.id);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        double best_cost = 1.7976931348623157e308;     // DBL_MAX, as the host starts
        int best_axis = -1, best_bin = -1;
        int lbox[6], rbox[6];
        uint32_t nL = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { lbox[k] = 0; rbox[k] = 0; }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            if (!__builtin_amdgcn_ballot_w64(ok3[ax])) continue;         // no node of this wave has an extent on this axis
            int L[6] = {kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax};
            int R[6] = {kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax};
            uint32_t lc = 0, rc = 0;
            const int bi = mybin[ax];
            for (uint32_t t = 0; t < n; ++t) {
                const uint4 a = rec[(nb + t) * 3], b = rec[(nb + t) * 3 + 1];
                const int bj = (int)((b.z >> (8 * ax)) & 0xFFu);
                const bool left = bj <= bi;
                const int v[6] = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y};
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int cl = k < 3 ? imin(L[k], v[k]) : imax(L[k], v[k]), cr = k < 3 ? imin(R[k], v[k]) : imax(R[k], v[k]);
                    L[k] = left ? cl : L[k];
                    R[k] = left ? R[k] : cr;
                }
                lc += left ? 1u : 0u;
                rc += left ? 0u : 1u;
            }
            double c = __builtin_inf();
            if (ok3[ax] && lc > 0u && rc > 0u) c = half_area(L) * (double)lc + half_area(R) * (double)rc;
            // the node's best candidate: minimum cost, ties to the lower bin (equal bins are the same partition)
            cand[lane] = SubCand{c, bi, 0};
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            double c2 = __builtin_inf();
            int bb = 0x7FFFFFFF, src = lane;
            for (uint32_t t = 0; t < n; ++t) {
                const SubCand q = cand[nb + t];
                if (q.cost < c2 || (q.cost == c2 && q.bin < bb)) { c2 = q.cost; bb = q.bin; src = (int)(nb + t); }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            int wl[6], wr[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { wl[k] = __shfl(L[k], src, 64); wr[k] = __shfl(R[k], src, 64); }
            const uint32_t wlc = __shfl(lc, src, 64);
            if (c2 < best_cost) {
                best_cost = c2; best_axis = ax; best_bin = bb; nL = wlc;
#pragma unroll
                for (int k = 0; k < 6; ++k) { lbox[k] = wl[k]; rbox[k] = wr[k]; }
            }
        }
        bool pred = false, have_split = false;
        if (best_axis >= 0) {
            have_split = fits(nL, (int)depth + 1, P) && fits(n - nL, (int)depth + 1, P);
            const int mb = best_axis == 0 ? mybin[0] : (best_axis == 1 ? mybin[1] : mybin[2]);
            pred = mb <= best_bin;
        }
        const bool need_median = live && !have_split;
        if (__builtin_amdgcn_ballot_w64(need_median)) {       // some node of the wave takes the median fallback
            const int ax = median_axis(cb);
            const float myc = ax == 0 ? pr.c[0] : (ax == 1 ? pr.c[1] : pr.c[2]);
            uint32_t rank = 0;
            for (uint32_t t = 0; t < n; ++t) {
                const uint4 c = rec[(nb + t) * 3 + 2];
                const float cj = __uint_as_float(ax == 0 ? c.x : (ax == 1 ? c.y : c.z));
                const uint32_t ij = rec[(nb + t) * 3 + 1].w;
                rank += key_less(cj, ij, myc, pr.id) ? 1u : 0u;
            }
            const uint32_t half = (n + 1u) / 2u;
            const bool mp = rank < half;
            rec[lane * 3 + 2].w = mp ? 1u : 0u;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            int ml[6] = {kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax};
            int mr[6] = {kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax};
            for (uint32_t t = 0; t < n; ++t) {
                const uint4 a = rec[(nb + t) * 3], b = rec[(nb + t) * 3 + 1];
                const bool left = rec[(nb + t) * 3 + 2].w != 0u;
                const int v[6] = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y};
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int cl = k < 3 ? imin(ml[k], v[k]) : imax(ml[k], v[k]), cr = k < 3 ? imin(mr[k], v[k]) : imax(mr[k], v[k]);
                    ml[k] = left ? cl : ml[k];
                    mr[k] = left ? mr[k] : cr;
                }
            }
            if (need_median) {
                nL = half; pred = mp;
#pragma unroll
                for (int k = 0; k < 6; ++k) { lbox[k] = ml[k]; rbox[k] = mr[k]; }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        // children: numbers for the inner ones (in position order over the wave's nodes), records written by each node's first lane
        const uint32_t lsz = nL, rsz = n - nL;
        const bool leader = live && (uint32_t)lane == nb;
        const bool linner = lsz > (uint32_t)P.max_leaf, rinner = rsz > (uint32_t)P.max_leaf;
        const unsigned long long mL = __builtin_amdgcn_ballot_w64(leader && linner), mR = __builtin_amdgcn_ballot_w64(leader && rinner);
        const uint32_t id_left = next_local + (uint32_t)__popcll(mL & below) + (uint32_t)__popcll(mR & below);
        const uint32_t id_right = id_left + (linner ? 1u : 0u);
        if (leader) {
            int32_t child[2];
            child[0] = linner ? (int32_t)id_left : ~(int32_t)((gbegin + nb) * 8u + lsz);
            child[1] = rinner ? (int32_t)id_right : ~(int32_t)((gbegin + nb + nL) * 8u + rsz);
            if (nid == 0) {
                rn.mid = gbegin + nb + nL;
                rn.child[0] = child[0]; rn.child[1] = child[1];
#pragma unroll
                for (int k = 0; k < 6; ++k) { rn.box[0][k] = lbox[k]; rn.box[1][k] = rbox[k]; }
            } else {
                SNode sn;
                sn.begin = gbegin + nb; sn.end = gbegin + ne; sn.mid = gbegin + nb + nL; sn.depth = depth;
                sn.child[0] = child[0]; sn.child[1] = child[1];
#pragma unroll
                for (int k = 0; k < 6; ++k) { sn.box[0][k] = lbox[k]; sn.box[1][k] = rbox[k]; }
                sn.pad[0] = sn.pad[1] = 0;
                stage[gbegin + nid - 1u] = sn;
            }
        }
        leaves += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(leader && !linner)) + (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(leader && !rinner));
        {
            const uint32_t lf = leader ? ((linner ? 0u : lsz) > (rinner ? 0u : rsz) ? (linner ? 0u : lsz) : (rinner ? 0u : rsz)) : 0u;
            maxl = (uint32_t)imax((int)maxl, wave_max((int)lf));
        }
        next_local += (uint32_t)__popcll(mL) + (uint32_t)__popcll(mR);
        // the two halves of every node change lanes (stable inside a side), then every POSITION learns its new node
        const unsigned long long nm = (ne >= 64u ? ~0ull : (1ull << ne) - 1ull) & ~((1ull << nb) - 1ull);
        const unsigned long long am = __builtin_amdgcn_ballot_w64(live), lm = __builtin_amdgcn_ballot_w64(live && pred);
        const uint32_t pos = !live ? (uint32_t)lane
                                   : (pred ? nb + (uint32_t)__popcll(lm & nm & below) : nb + nL + (uint32_t)__popcll(am & ~lm & nm & below));
        {
            const int dst = (int)pos * 4;
#define LRC_MOVE_F(x) x = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(x)))
            LRC_MOVE_F(pr.lo[0]); LRC_MOVE_F(pr.lo[1]); LRC_MOVE_F(pr.lo[2]);
            LRC_MOVE_F(pr.hi[0]); LRC_MOVE_F(pr.hi[1]); LRC_MOVE_F(pr.hi[2]);
            LRC_MOVE_F(pr.c[0]); LRC_MOVE_F(pr.c[1]); LRC_MOVE_F(pr.c[2]);
#undef LRC_MOVE_F
            pr.id = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)pr.id);
        }
        const uint32_t idl = (uint32_t)__shfl((int)id_left, (int)nb, 64), idr = (uint32_t)__shfl((int)id_right, (int)nb, 64);
        if (live) {
            const uint32_t mid = nb + nL;
            if ((uint32_t)lane < mid) { ne = mid; nid = idl; live = linner; }
            else { nb = mid; nid = idr; live = rinner; }
        }
        ++depth;
    }
    if ((uint32_t)lane < n_root) P.final_id[gbegin + lane] = pr.id;
    if (lane == 0) info[w] = SubInfo{next_local - 1u, leaves, maxl, depth};
}

// one workgroup: exclusive scan of the per-root node counts (bases[r]) and the totals of the subtree pass.  Every thread owns
// a contiguous run of roots: one pass for its sum, a scan of the 1024 sums (wave shuffles + 16 wave totals), one pass to write.
__global__ __launch_bounds__(1024) void k_sub_scan(const SubInfo* info, uint32_t n_roots, uint32_t* bases, SubTotals* totals) {
    __shared__ uint32_t s_wave[16], s_leaves, s_maxl, s_maxd;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) { s_leaves = 0; s_maxl = 0; s_maxd = 0; }
    __syncthreads();
    const uint32_t per = (n_roots + 1023u) / 1024u, first = tid * per, last = first + per < n_roots ? first + per : n_roots;
    uint32_t sum = 0, lv = 0, ml = 0, md = 0;
    for (uint32_t i = first; i < last; ++i) {
        const SubInfo f = info[i];
        sum += f.count; lv += f.leaves; ml = ml > f.max_leaf ? ml : f.max_leaf; md = md > f.max_depth ? md : f.max_depth;
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    atomicAdd(&s_leaves, lv); atomicMax(&s_maxl, ml); atomicMax(&s_maxd, md);
    __syncthreads();
    uint32_t base = incl - sum, total = 0;
    for (uint32_t k = 0; k < 16; ++k) { if (k < wave) base += s_wave[k]; total += s_wave[k]; }
    for (uint32_t i = first; i < last; ++i) { bases[i] = base; base += info[i].count; }
    if (tid == 0) *totals = SubTotals{total, s_leaves, s_maxl, s_maxd};
}

// one wave per root: its staged nodes to their numbers first + bases[r] + j, wave-local child numbers made global
__global__ __launch_bounds__(256) void k_sub_place(TNode* nodes, const SubRoot* roots, uint32_t n_roots, const SNode* stage,
                                                   const SubInfo* info, const uint32_t* bases, uint32_t first) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_roots) return;
    const uint32_t lane = threadIdx.x & 63u;
    const SubRoot root = roots[w];
    const uint32_t cnt = info[w].count, at = first + bases[w];
    TNode& rn = nodes[root.g];
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (rn.child[c] > 0) rn.child[c] = (int32_t)(at + (uint32_t)rn.child[c] - 1u);
    }
    if (lane < cnt) {
        const SNode sn = stage[rn.begin + lane];
        TNode nd;
        nd.begin = sn.begin; nd.end = sn.end; nd.mid = sn.mid; nd.depth = sn.depth;
#pragma unroll
        for (int c = 0; c < 2; ++c) nd.child[c] = sn.child[c] > 0 ? (int32_t)(at + (uint32_t)sn.child[c] - 1u) : sn.child[c];
        nd.work = -1; nd.cb_valid = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { nd.cb[k] = 0; nd.box[0][k] = sn.box[0][k]; nd.box[1][k] = sn.box[1][k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) nd.pad[k] = 0;
        nodes[at + lane] = nd;
    }
}

// ---- nodes of <= 16 primitives: four to a wave ---------------------------------------------------------------------------
// Lane i of a 16-lane group holds primitive i of its node and evaluates ONE candidate: the split after the bin its own
// centroid falls into.  The host's sweep over the 64 bins changes the partition only at occupied bins, equal partitions have
// equal costs and the first minimum wins, so the minimum over the occupied bins with ties to the lower bin is the host's
// choice.  The group's primitives sit in LDS (8 words each: box as ordered ints, the three bin numbers, the row); every lane
// walks the 16 records (broadcast reads) and grows the left or the right side of its candidate.
DI int group_min(int v) {
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) v = imin(v, __shfl_xor(v, d, 64));
    return v;
}
DI int group_max(int v) {
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) v = imax(v, __shfl_xor(v, d, 64));
    return v;
}

__global__ __launch_bounds__(256) void k_tiny(const Params P, const uint32_t* list, uint32_t n_list) {
    __shared__ uint4 s_rec[4][4 * kTinyMax * 2];       // per wave: 4 groups x 16 primitives x 2 x uint4
    const uint32_t wave0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;       // first node of this wave
    if (wave0 >= n_list) return;                    // whole waves leave; no workgroup barrier below
    const int lane = threadIdx.x & 63, gl = lane & 15, g0 = lane & 48;
    uint4* rec = s_rec[threadIdx.x >> 6] + (size_t)(g0 >> 4) * kTinyMax * 2;
    const uint32_t w = wave0 + (uint32_t)(g0 >> 4);
    const bool node_ok = w < n_list;
    const uint32_t g = node_ok ? list[w] : 0u;
    TNode& nd = P.nodes[g];
    const uint32_t begin = node_ok ? nd.begin : 0u, n = node_ok ? nd.end - nd.begin : 0u;
    const int depth = (int)nd.depth;
    const bool act = (uint32_t)gl < n;
    LoadedPrim pr{};
    if (act) pr = load_prim(P.cur + begin + gl);
    int cb[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cb[k] = group_min(act ? enc(pr.c[k]) : kEncPosMax);
        cb[3 + k] = group_max(act ? enc(pr.c[k]) : kEncNegMax);
    }
    int mybin[3] = {0, 0, 0};
    bool ok3[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float lo = dec(cb[ax]), ext = dec(cb[3 + ax]) - lo;
        ok3[ax] = (ext > 0.0f) && !P.median_only;
        mybin[ax] = ok3[ax] ? bin_of(pr.c[ax], lo, (float)kBinsN / ext) : 0;
    }
    rec[gl * 2] = make_uint4((uint32_t)enc(pr.lo[0]), (uint32_t)enc(pr.lo[1]), (uint32_t)enc(pr.lo[2]), (uint32_t)enc(pr.hi[0]));
    rec[gl * 2 + 1] = make_uint4((uint32_t)enc(pr.hi[1]), (uint32_t)enc(pr.hi[2]),
                                 (uint32_t)mybin[0] | ((uint32_t)mybin[1] << 8) | ((uint32_t)mybin[2] << 16) | (act ? 1u << 24 : 0u), pr.id);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    double best_cost = 1.7976931348623157e308;
    int best_axis = -1, best_bin = -1;
    int lbox[6], rbox[6];
    uint32_t nL = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { lbox[k] = 0; rbox[k] = 0; }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        if (!__builtin_amdgcn_ballot_w64(ok3[ax])) continue;         // no group of this wave has an extent on this axis
        int L[6] = {kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax};
        int R[6] = {kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax};
        uint32_t lc = 0, rc = 0;
        const int bi = mybin[ax];
#pragma unroll 4
        for (int j = 0; j < kTinyMax; ++j) {
            const uint4 a = rec[j * 2], b = rec[j * 2 + 1];
            const bool valid = (b.z >> 24) != 0u;
            const int bj = (int)((b.z >> (8 * ax)) & 0xFFu);
            const bool left = valid && bj <= bi, right = valid && bj > bi;
            const int v[6] = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int cl = k < 3 ? imin(L[k], v[k]) : imax(L[k], v[k]), cr = k < 3 ? imin(R[k], v[k]) : imax(R[k], v[k]);
                L[k] = left ? cl : L[k];
                R[k] = right ? cr : R[k];
            }
            lc += left ? 1u : 0u;
            rc += right ? 1u : 0u;
        }
        double c = __builtin_inf();
        if (act && ok3[ax] && lc > 0u && rc > 0u) c = half_area(L) * (double)lc + half_area(R) * (double)rc;
        int bb = bi, src = lane;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {
            const double oc = __shfl_xor(c, d, 64);
            const int ob = __shfl_xor(bb, d, 64), os = __shfl_xor(src, d, 64);
            if (oc < c || (oc == c && ob < bb)) { c = oc; bb = ob; src = os; }
        }
        // the winner's two sides (every lane of the group reads them from the winning lane)
        int wl[6], wr[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { wl[k] = __shfl(L[k], src, 64); wr[k] = __shfl(R[k], src, 64); }
        const uint32_t wlc = __shfl(lc, src, 64);
        if (c < best_cost) {
            best_cost = c; best_axis = ax; best_bin = bb; nL = wlc;
#pragma unroll
            for (int k = 0; k < 6; ++k) { lbox[k] = wl[k]; rbox[k] = wr[k]; }
        }
    }
    bool pred = false, have_split = false;
    if (best_axis >= 0) {
        have_split = fits(nL, depth + 1, P) && fits(n - nL, depth + 1, P);
        const int mb = best_axis == 0 ? mybin[0] : (best_axis == 1 ? mybin[1] : mybin[2]);
        pred = mb <= best_bin;
    }
    const bool need_median = node_ok && !have_split;
    if (__builtin_amdgcn_ballot_w64(need_median)) {       // some group of the wave takes the median fallback
        const int ax = median_axis(cb);
        const float myc = ax == 0 ? pr.c[0] : (ax == 1 ? pr.c[1] : pr.c[2]);
        uint32_t rank = 0;
        for (int j = 0; j < kTinyMax; ++j) {
            const float cj = __shfl(myc, g0 + j, 64);
            const uint32_t ij = __shfl(pr.id, g0 + j, 64);
            rank += ((uint32_t)j < n && key_less(cj, ij, myc, pr.id)) ? 1u : 0u;
        }
        const uint32_t half = (n + 1u) / 2u;
        const bool mp = rank < half;
        int ml[6], mr[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ml[k] = group_min(act && mp ? enc(pr.lo[k]) : kEncPosMax);
            ml[3 + k] = group_max(act && mp ? enc(pr.hi[k]) : kEncNegMax);
            mr[k] = group_min(act && !mp ? enc(pr.lo[k]) : kEncPosMax);
            mr[3 + k] = group_max(act && !mp ? enc(pr.hi[k]) : kEncNegMax);
        }
        if (need_median) {
            nL = half; pred = mp;
#pragma unroll
            for (int k = 0; k < 6; ++k) { lbox[k] = ml[k]; rbox[k] = mr[k]; }
        }
    }
    const unsigned long long am = __ballot(act), lm = __ballot(act && pred);
    const uint32_t gam = (uint32_t)(am >> g0) & 0xFFFFu, glm = (uint32_t)(lm >> g0) & 0xFFFFu;
    const uint32_t below = (1u << gl) - 1u;
    if (act) {
        const uint32_t pos = pred ? (uint32_t)__popc(glm & below) : nL + (uint32_t)__popc(gam & ~glm & below);
        store_prim(P.nxt + begin + pos, pr);
        const uint32_t cn = pred ? nL : n - nL;
        if (cn <= (uint32_t)P.max_leaf) P.final_id[begin + pos] = pr.id;
    }
    if (gl == 0 && node_ok) {
        nd.mid = begin + nL;
#pragma unroll
        for (int k = 0; k < 6; ++k) { nd.box[0][k] = lbox[k]; nd.box[1][k] = rbox[k]; }
    }
}

// ---- nodes of 65 .. 1024 primitives: one workgroup each -----------------------------------------------------------------
struct BestRec { double cost; int bin; uint32_t lcnt; int lbox[6], rbox[6]; };

__global__ __launch_bounds__(256) void k_medium(const Params P, const uint32_t* list) {
    __shared__ int s_bins[3][7 * kBinsN];
    __shared__ int s_cb[6];
    __shared__ BestRec s_best[3];
    __shared__ int s_box[2][6];
    __shared__ uint32_t s_cur[2];
    __shared__ float s_keyc[kMediumMax];
    __shared__ uint32_t s_keyi[kMediumMax];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = list[blockIdx.x];
    TNode& nd = P.nodes[g];
    const uint32_t begin = nd.begin, n = nd.end - begin;
    const int depth = (int)nd.depth;
    LoadedPrim pr[4];
    bool act[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = (uint32_t)tid + (uint32_t)k * 256u;
        act[k] = i < n;
        pr[k] = LoadedPrim{};
        if (act[k]) pr[k] = load_prim(P.cur + begin + i);
    }
    if (tid < 6) s_cb[tid] = tid < 3 ? kEncPosMax : kEncNegMax;
    if (tid < 2) s_cur[tid] = 0u;
    if (tid < 12) s_box[tid / 6][tid % 6] = (tid % 6) < 3 ? kEncPosMax : kEncNegMax;
    for (int i = tid; i < 3 * 7 * kBinsN; i += 256) {
        const int k = (i / kBinsN) % 7;
        (&s_bins[0][0])[i] = k < 3 ? kEncPosMax : (k < 6 ? kEncNegMax : 0);
    }
    __syncthreads();
    {
        int m[6] = {kEncPosMax, kEncPosMax, kEncPosMax, kEncNegMax, kEncNegMax, kEncNegMax};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (act[k]) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { const int e = enc(pr[k].c[a]); m[a] = imin(m[a], e); m[3 + a] = imax(m[3 + a], e); }
            }
#pragma unroll
        for (int a = 0; a < 3; ++a) { m[a] = wave_min(m[a]); m[3 + a] = wave_max(m[3 + a]); }
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { atomicMin(&s_cb[a], m[a]); atomicMax(&s_cb[3 + a], m[3 + a]); }
        }
    }
    __syncthreads();
    int cb[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cb[k] = s_cb[k];
    float lo3[3], sc3[3];
    bool ok3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo3[a] = dec(cb[a]);
        const float ext = dec(cb[3 + a]) - lo3[a];
        ok3[a] = (ext > 0.0f) && !P.median_only;
        sc3[a] = ok3[a] ? (float)kBinsN / ext : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (act[k]) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (ok3[a]) {
                    const int b = bin_of(pr[k].c[a], lo3[a], sc3[a]);
                    int* bins = s_bins[a];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        atomicMin(&bins[q * kBinsN + b], enc(pr[k].lo[q]));
                        atomicMax(&bins[(3 + q) * kBinsN + b], enc(pr[k].hi[q]));
                    }
                    atomicAdd(&bins[6 * kBinsN + b], 1);
                }
        }
    __syncthreads();
    if (wave < 3) {
        BestRec br;
        br.cost = __builtin_inf(); br.bin = -1; br.lcnt = 0;
        const bool ok = wave == 0 ? ok3[0] : (wave == 1 ? ok3[1] : ok3[2]);
        if (ok) {
            int bb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) bb[k] = s_bins[wave][k * kBinsN + lane];
            const AxisBest r = eval_axis(lane, bb, (uint32_t)s_bins[wave][6 * kBinsN + lane]);
            br.cost = r.cost; br.bin = r.bin; br.lcnt = r.lcnt;
#pragma unroll
            for (int k = 0; k < 6; ++k) { br.lbox[k] = r.lbox[k]; br.rbox[k] = r.rbox[k]; }
        }
        if (lane == 0) s_best[wave] = br;
    }
    __syncthreads();
    double best_cost = 1.7976931348623157e308;
    int best_axis = -1;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (s_best[a].cost < best_cost) { best_cost = s_best[a].cost; best_axis = a; }
    uint32_t nL = 0;
    bool have_split = false;
    int best_bin = 0;
    if (best_axis >= 0) {
        nL = s_best[best_axis].lcnt;
        best_bin = s_best[best_axis].bin;
        have_split = fits(nL, depth + 1, P) && fits(n - nL, depth + 1, P);
    }
    bool pred[4] = {false, false, false, false};
    if (have_split) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c = best_axis == 0 ? pr[k].c[0] : (best_axis == 1 ? pr[k].c[1] : pr[k].c[2]);
            const float lo = best_axis == 0 ? lo3[0] : (best_axis == 1 ? lo3[1] : lo3[2]);
            const float sc = best_axis == 0 ? sc3[0] : (best_axis == 1 ? sc3[1] : sc3[2]);
            pred[k] = bin_of(c, lo, sc) <= best_bin;
        }
    } else {
        const int ax = median_axis(cb);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (act[k]) {
                const uint32_t i = (uint32_t)tid + (uint32_t)k * 256u;
                s_keyc[i] = ax == 0 ? pr[k].c[0] : (ax == 1 ? pr[k].c[1] : pr[k].c[2]);
                s_keyi[i] = pr[k].id;
            }
        __syncthreads();
        nL = (n + 1u) / 2u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float myc = ax == 0 ? pr[k].c[0] : (ax == 1 ? pr[k].c[1] : pr[k].c[2]);
            uint32_t rank = 0;
            if (act[k])
                for (uint32_t j = 0; j < n; ++j) rank += key_less(s_keyc[j], s_keyi[j], myc, pr[k].id) ? 1u : 0u;
            pred[k] = rank < nL;
        }
        // child boxes = exact bounds of the two halves
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (act[k]) {
                int* bx = s_box[pred[k] ? 0 : 1];
#pragma unroll
                for (int q = 0; q < 3; ++q) { atomicMin(&bx[q], enc(pr[k].lo[q])); atomicMax(&bx[3 + q], enc(pr[k].hi[q])); }
            }
    }
    // partition into the other buffer: positions from two LDS cursors (the order inside a side is immaterial)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long am = __ballot(act[k]), lm = __ballot(act[k] && pred[k]);
        const uint32_t cl = (uint32_t)__popcll(lm), cr = (uint32_t)__popcll(am & ~lm);
        uint32_t bl = 0, brr = 0;
        if (lane == 0) { bl = cl ? atomicAdd(&s_cur[0], cl) : 0u; brr = cr ? atomicAdd(&s_cur[1], cr) : 0u; }
        bl = __shfl(bl, 0, 64); brr = __shfl(brr, 0, 64);
        if (act[k]) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t pos = pred[k] ? bl + (uint32_t)__popcll(lm & below)
                                         : nL + brr + (uint32_t)__popcll(am & ~lm & below);
            store_prim(P.nxt + begin + pos, pr[k]);
            const uint32_t cn = pred[k] ? nL : n - nL;
            if (cn <= (uint32_t)P.max_leaf) P.final_id[begin + pos] = pr[k].id;
        }
    }
    __syncthreads();
    if (tid == 0) {
        nd.mid = begin + nL;
        if (have_split) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { nd.box[0][k] = s_best[best_axis].lbox[k]; nd.box[1][k] = s_best[best_axis].rbox[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) { nd.box[0][k] = s_box[0][k]; nd.box[1][k] = s_box[1][k]; }
        }
    }
}

// ---- nodes of > 1024 primitives: workgroups per chunk, bins combined in HBM ------------------------------------------
// chunk_start[i] = first chunk of big node i (exclusive prefix), chunk_start[nb] = number of chunks
__global__ __launch_bounds__(1024) void k_big_prefix(TNode* nodes, const uint32_t* list, uint32_t nb, BigWork* work,
                                                     uint32_t* chunk_start) {
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + tid;
        uint32_t v = 0;
        if (i < nb) {
            const uint32_t g = list[i];
            v = (nodes[g].end - nodes[g].begin + kChunk - 1) / kChunk;
            nodes[g].work = (int32_t)i;
            work[i].g = g;
        }
        s_part[tid] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t add = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += add;
            __syncthreads();
        }
        const uint32_t carry = s_carry;
        if (i < nb) chunk_start[i] = carry + s_part[tid] - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + s_part[1023];
        __syncthreads();
    }
    if (tid == 0) chunk_start[nb] = s_carry;
}

__global__ __launch_bounds__(256) void k_big_bins_init(int* gbins, uint32_t total) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int k = (i / kBinsN) % 7;
    gbins[i] = k < 3 ? kEncPosMax : (k < 6 ? kEncNegMax : 0);
}

DI bool chunk_of(const uint32_t* chunk_start, uint32_t nb, uint32_t b, uint32_t& slot, uint32_t& c) {
    if (b >= chunk_start[nb]) return false;
    uint32_t lo = 0, hi = nb;                 // largest slot with chunk_start[slot] <= b
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_start[mid] <= b) lo = mid; else hi = mid;
    }
    slot = lo;
    c = b - chunk_start[lo];
    return true;
}

__global__ __launch_bounds__(256) void k_big_bin(const Params P, const BigWork* work, const uint32_t* chunk_start,
                                                 uint32_t nb, int* gbins) {
    __shared__ int s_bins[3][7 * kBinsN];
    uint32_t slot, c;
    if (!chunk_of(chunk_start, nb, blockIdx.x, slot, c)) return;
    const int tid = threadIdx.x;
    const TNode& nd = P.nodes[work[slot].g];
    const uint32_t first = nd.begin + c * kChunk, last = min(nd.end, first + kChunk);
    float lo3[3], sc3[3];
    bool ok3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo3[a] = dec(nd.cb[a]);
        const float ext = dec(nd.cb[3 + a]) - lo3[a];
        ok3[a] = ext > 0.0f;
        sc3[a] = ok3[a] ? (float)kBinsN / ext : 0.0f;
    }
    for (int i = tid; i < 3 * 7 * kBinsN; i += 256) {
        const int k = (i / kBinsN) % 7;
        (&s_bins[0][0])[i] = k < 3 ? kEncPosMax : (k < 6 ? kEncNegMax : 0);
    }
    __syncthreads();
    for (uint32_t i = first + tid; i < last; i += 256) {
        const LoadedPrim pr = load_prim(P.cur + i);
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (ok3[a]) {
                const int b = bin_of(pr.c[a], lo3[a], sc3[a]);
                int* bins = s_bins[a];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    atomicMin(&bins[q * kBinsN + b], enc(pr.lo[q]));
                    atomicMax(&bins[(3 + q) * kBinsN + b], enc(pr.hi[q]));
                }
                atomicAdd(&bins[6 * kBinsN + b], 1);
            }
    }
    __syncthreads();
    if (tid < 3 * kBinsN) {
        const int a = tid / kBinsN, b = tid % kBinsN;
        const int cnt = s_bins[a][6 * kBinsN + b];
        if (cnt > 0) {
            int* gb = gbins + ((size_t)slot * 3 + a) * 7 * kBinsN;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                atomicMin(&gb[q * kBinsN + b], s_bins[a][q * kBinsN + b]);
                atomicMax(&gb[(3 + q) * kBinsN + b], s_bins[a][(3 + q) * kBinsN + b]);
            }
            atomicAdd(&gb[6 * kBinsN + b], cnt);
        }
    }
}

// The bins are left EMPTY for the next level: this workgroup clears the slot it has read and slot + gridDim.x (a level has
// at most twice as many big nodes as the one above it, so slots [0, 2 nb) cover it; `slots_cap` bounds the array) -- the
// k_big_bins_init launch per level is gone, only the first big level still needs it.
__global__ __launch_bounds__(192) void k_big_eval(const Params P, BigWork* work, int* gbins, LevelCounters* lc,
                                                  uint32_t* median_list, uint32_t slots_cap) {
    __shared__ BestRec s_best[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t slot = blockIdx.x;
    BigWork& w = work[slot];
    TNode& nd = P.nodes[w.g];
    const uint32_t n = nd.end - nd.begin;
    const int depth = (int)nd.depth;
    const float lo = dec(nd.cb[wave]), ext = dec(nd.cb[3 + wave]) - lo;
    BestRec br;
    br.cost = __builtin_inf(); br.bin = -1; br.lcnt = 0;
    if (ext > 0.0f && !P.median_only) {
        const int* gb = gbins + ((size_t)slot * 3 + wave) * 7 * kBinsN;
        int bb[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) bb[k] = gb[k * kBinsN + lane];
        const AxisBest r = eval_axis(lane, bb, (uint32_t)gb[6 * kBinsN + lane]);
        br.cost = r.cost; br.bin = r.bin; br.lcnt = r.lcnt;
#pragma unroll
        for (int k = 0; k < 6; ++k) { br.lbox[k] = r.lbox[k]; br.rbox[k] = r.rbox[k]; }
    }
    if (lane == 0) s_best[wave] = br;
    __syncthreads();                     // every wave has read its axis' bins
    for (uint32_t which = 0; which < 2; ++which) {
        const uint32_t sl = slot + which * gridDim.x;
        if (sl >= slots_cap) break;
        int* gb = gbins + (size_t)sl * 3 * 7 * kBinsN;
        for (int i = tid; i < 3 * 7 * kBinsN; i += 192) {
            const int k = (i / kBinsN) % 7;
            gb[i] = k < 3 ? kEncPosMax : (k < 6 ? kEncNegMax : 0);
        }
    }
    if (tid != 0) return;
    double best_cost = 1.7976931348623157e308;
    int best_axis = -1;
    for (int a = 0; a < 3; ++a)
        if (s_best[a].cost < best_cost) { best_cost = s_best[a].cost; best_axis = a; }
    bool have_split = false;
    uint32_t nL = 0;
    if (best_axis >= 0) {
        nL = s_best[best_axis].lcnt;
        have_split = fits(nL, depth + 1, P) && fits(n - nL, depth + 1, P);
    }
    w.cur_left = 0; w.cur_right = 0;
    for (int k = 0; k < 6; ++k) { w.ccb[0][k] = w.ccb[1][k] = k < 3 ? kEncPosMax : kEncNegMax; }
    if (have_split) {
        w.median = 0;
        w.axis = best_axis; w.bin = s_best[best_axis].bin;
        w.lo = dec(nd.cb[best_axis]);
        w.scale = (float)kBinsN / (dec(nd.cb[3 + best_axis]) - w.lo);
        for (int k = 0; k < 6; ++k) { nd.box[0][k] = s_best[best_axis].lbox[k]; nd.box[1][k] = s_best[best_axis].rbox[k]; }
    } else {
        w.median = 1;
        w.axis = median_axis(nd.cb);
        nL = (n + 1u) / 2u;
        for (int k = 0; k < 6; ++k) { nd.box[0][k] = nd.box[1][k] = k < 3 ? kEncPosMax : kEncNegMax; }
        median_list[atomicAdd(&lc->n_median, 1u)] = slot;
    }
    nd.mid = nd.begin + nL;
}

__global__ __launch_bounds__(256) void k_big_scatter(const Params P, BigWork* work, const uint32_t* chunk_start,
                                                     uint32_t nb) {
    __shared__ uint32_t s_cl[16], s_cr[16], s_base[2];
    __shared__ int s_ccb[2][6];
    uint32_t slot, c;
    if (!chunk_of(chunk_start, nb, blockIdx.x, slot, c)) return;
    BigWork& w = work[slot];
    if (w.median) return;                       // split by the sort path instead
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TNode& nd = P.nodes[w.g];
    const uint32_t first = nd.begin + c * kChunk, last = min(nd.end, first + kChunk);
    const uint32_t nL = nd.mid - nd.begin, nR = nd.end - nd.mid;
    const int ax = w.axis, bin = w.bin;
    const float lo = w.lo, sc = w.scale;
    if (tid < 12) s_ccb[tid / 6][tid % 6] = (tid % 6) < 3 ? kEncPosMax : kEncNegMax;
    LoadedPrim pr[4];
    bool act[4], pred[4];
    unsigned long long am[4], lm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = first + (uint32_t)tid + (uint32_t)k * 256u;
        act[k] = i < last;
        pr[k] = LoadedPrim{};
        if (act[k]) pr[k] = load_prim(P.cur + i);
        const float cc = ax == 0 ? pr[k].c[0] : (ax == 1 ? pr[k].c[1] : pr[k].c[2]);
        pred[k] = act[k] && bin_of(cc, lo, sc) <= bin;
        am[k] = __ballot(act[k]); lm[k] = __ballot(pred[k]);
        if (lane == 0) { s_cl[k * 4 + wave] = (uint32_t)__popcll(lm[k]); s_cr[k * 4 + wave] = (uint32_t)__popcll(am[k] & ~lm[k]); }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t tl = 0, tr = 0;
        for (int q = 0; q < 16; ++q) { const uint32_t a = s_cl[q], b = s_cr[q]; s_cl[q] = tl; s_cr[q] = tr; tl += a; tr += b; }
        s_base[0] = tl ? atomicAdd(&w.cur_left, tl) : 0u;
        s_base[1] = tr ? atomicAdd(&w.cur_right, tr) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (act[k]) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t pos = pred[k] ? s_base[0] + s_cl[k * 4 + wave] + (uint32_t)__popcll(lm[k] & below)
                                         : nL + s_base[1] + s_cr[k * 4 + wave] + (uint32_t)__popcll(am[k] & ~lm[k] & below);
            store_prim(P.nxt + nd.begin + pos, pr[k]);
            if ((pred[k] ? nL : nR) <= (uint32_t)P.max_leaf) P.final_id[nd.begin + pos] = pr[k].id;
            int* cc = s_ccb[pred[k] ? 0 : 1];
#pragma unroll
            for (int q = 0; q < 3; ++q) { const int e = enc(pr[k].c[q]); atomicMin(&cc[q], e); atomicMax(&cc[3 + q], e); }
        }
    __syncthreads();
    if (tid < 12) {
        const int ch = tid / 6, q = tid % 6;
        if (q < 3) atomicMin(&w.ccb[ch][q], s_ccb[ch][q]); else atomicMax(&w.ccb[ch][q], s_ccb[ch][q]);
    }
}

// big-node median fallback: keys (centroid along the axis, row) in sortable form; the sorted order IS the partition
__global__ __launch_bounds__(256) void k_median_keys(const PrimRec* cur, uint32_t begin, uint32_t n, int ax,
                                                     uint64_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const LoadedPrim pr = load_prim(cur + begin + i);
    float c = ax == 0 ? pr.c[0] : (ax == 1 ? pr.c[1] : pr.c[2]);
    if (c == 0.0f) c = 0.0f;                    // -0.0 and +0.0 compare equal on the host: one key
    const uint32_t u = (uint32_t)enc(c) ^ 0x80000000u;      // order-preserving, unsigned
    keys[i] = ((uint64_t)u << 32) | pr.id;
    vals[i] = i;
}

__global__ __launch_bounds__(256) void k_median_gather(const Params P, BigWork* w, uint32_t n, const uint32_t* order) {
    __shared__ int s_box[2][6], s_ccb[2][6];
    const int tid = threadIdx.x;
    TNode& nd = P.nodes[w->g];
    const uint32_t half = nd.mid - nd.begin;
    if (tid < 12) { s_box[tid / 6][tid % 6] = s_ccb[tid / 6][tid % 6] = (tid % 6) < 3 ? kEncPosMax : kEncNegMax; }
    __syncthreads();
    const uint32_t r = blockIdx.x * 256 + tid;
    if (r < n) {
        const LoadedPrim pr = load_prim(P.cur + nd.begin + order[r]);
        store_prim(P.nxt + nd.begin + r, pr);
        const int ch = r < half ? 0 : 1;
        if ((ch == 0 ? half : n - half) <= (uint32_t)P.max_leaf) P.final_id[nd.begin + r] = pr.id;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            atomicMin(&s_box[ch][q], enc(pr.lo[q])); atomicMax(&s_box[ch][3 + q], enc(pr.hi[q]));
            const int e = enc(pr.c[q]);
            atomicMin(&s_ccb[ch][q], e); atomicMax(&s_ccb[ch][3 + q], e);
        }
    }
    __syncthreads();
    if (tid < 12) {
        const int ch = tid / 6, q = tid % 6;
        if (q < 3) { atomicMin(&nd.box[ch][q], s_box[ch][q]); atomicMin(&w->ccb[ch][q], s_ccb[ch][q]); }
        else { atomicMax(&nd.box[ch][q], s_box[ch][q]); atomicMax(&w->ccb[ch][q], s_ccb[ch][q]); }
    }
}

// ---- children of a level: breadth-first numbers (ordered scan), work lists of the next level ---------------------------
__global__ __launch_bounds__(256) void k_emit_count(const TNode* nodes, uint32_t base, uint32_t n, int max_leaf,
                                                    uint32_t* partial) {
    __shared__ uint32_t s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t cnt = 0;
    if (i < n) {
        const TNode& nd = nodes[base + i];
        if (nd.work != -2)       // a subtree root (k_subtree builds what is below it)
            cnt = ((nd.mid - nd.begin) > (uint32_t)max_leaf ? 1u : 0u) + ((nd.end - nd.mid) > (uint32_t)max_leaf ? 1u : 0u);
    }
    if (cnt) atomicAdd(&s_sum, cnt);
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_sum;
}

__global__ __launch_bounds__(256) void k_emit_write(TNode* nodes, uint32_t base, uint32_t n, uint32_t next_base,
                                                    int max_leaf, const uint32_t* partial, const BigWork* work,
                                                    uint32_t* list_tiny, uint32_t* list_small, uint32_t* list_medium,
                                                    uint32_t* list_big, LevelCounters* next, int sub_mode, SubRoot* sub_roots,
                                                    uint32_t* n_sub_total, uint32_t child_parity) {
    __shared__ uint32_t s_red[256];
    __shared__ uint32_t s_wave[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // inner children in front of this workgroup
    uint32_t acc = 0;
    for (uint32_t j = tid; j < blockIdx.x; j += 256) acc += partial[j];
    s_red[tid] = acc;
    __syncthreads();
    for (uint32_t off = 128; off >= 1; off >>= 1) {
        if (tid < off) s_red[tid] += s_red[tid + off];
        __syncthreads();
    }
    const uint32_t block_base = s_red[0];
    const uint32_t i = blockIdx.x * 256 + tid;
    uint32_t nl = 0, nr = 0, cnt = 0;
    TNode* nd = nullptr;
    if (i < n) {
        nd = nodes + base + i;
        if (nd->work == -2) nd = nullptr;          // a subtree root: no children here
        else {
            nl = nd->mid - nd->begin; nr = nd->end - nd->mid;
            cnt = (nl > (uint32_t)max_leaf ? 1u : 0u) + (nr > (uint32_t)max_leaf ? 1u : 0u);
        }
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t k = 0; k < wave; ++k) wbase += s_wave[k];
    uint32_t off = block_base + wbase + incl - cnt;
    if (blockIdx.x == gridDim.x - 1 && tid == 255) next->n_nodes = block_base + wbase + incl;
    if (!nd) return;
    uint32_t leaves = 0, maxl = 0;
    for (int c = 0; c < 2; ++c) {
        const uint32_t b = c == 0 ? nd->begin : nd->mid, e = c == 0 ? nd->mid : nd->end, m = e - b;
        if (m > (uint32_t)max_leaf) {
            const uint32_t g = next_base + off;
            ++off;
            nd->child[c] = (int32_t)g;
            TNode& ch = nodes[g];
            ch.begin = b; ch.end = e; ch.mid = b; ch.depth = nd->depth + 1u;
            ch.child[0] = ch.child[1] = 0;
            ch.work = -1;
            ch.cb_valid = 0;
            if (nd->work >= 0) {               // a big parent's scatter has reduced the children's centroid bounds
                ch.cb_valid = 1;
                for (int k = 0; k < 6; ++k) ch.cb[k] = work[nd->work].ccb[c][k];
            }
            if (sub_mode && m <= (uint32_t)kSmallMax) {
                ch.work = -2;
                sub_roots[atomicAdd(n_sub_total, 1u)] = SubRoot{g, child_parity};
                atomicAdd(&next->n_sub, 1u);
            }
            else if (m <= (uint32_t)kTinyMax) list_tiny[atomicAdd(&next->n_tiny, 1u)] = g;
            else if (m <= (uint32_t)kSmallMax) list_small[atomicAdd(&next->n_small, 1u)] = g;
            else if (m <= (uint32_t)kMediumMax) list_medium[atomicAdd(&next->n_medium, 1u)] = g;
            else list_big[atomicAdd(&next->n_big, 1u)] = g;
        } else {
            nd->child[c] = ~(int32_t)(b * 8u + m);
            ++leaves;
            maxl = m > maxl ? m : maxl;
        }
    }
    if (leaves) { atomicAdd(&next->n_leaves, leaves); atomicMax(&next->max_leaf, maxl); }
}

// big children of a big parent need valid centroid bounds; children of small / medium parents that are big cannot exist
// (a child is never larger than its parent), and medium / small kernels reduce their own bounds

// ---- final layout -----------------------------------------------------------------------------------------------------
// tail nodes (breadth-first index >= head): rank by (group, begin, depth) = the host's depth-first order of the queue
// entries at the cut (bvh_build.cpp "relayout"): entries of the cut level that were not popped come first (begin >= B0)
__global__ __launch_bounds__(256) void k_tail_keys(const TNode* nodes, uint32_t head, uint32_t nn, uint64_t* keys,
                                                   uint32_t* vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (head + i >= nn) return;
    const uint32_t b0 = nodes[head].begin;
    const TNode& nd = nodes[head + i];
    keys[i] = ((uint64_t)(nd.begin >= b0 ? 0u : 1u) << 40) | ((uint64_t)nd.begin << 6) | (uint64_t)(nd.depth & 63u);
    vals[i] = head + i;
}

__global__ __launch_bounds__(256) void k_new_index(uint32_t head, uint32_t nn, const uint32_t* sorted_vals,
                                                   uint32_t* new_of_old) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    if (i < head) new_of_old[i] = i;
    else new_of_old[sorted_vals[i - head]] = i;
}

struct QGridDev { double bd[3], invWd[3], cell[3]; int enabled; };

DI bool qbox_dev(const QGridDev& g, const float* lo, const float* hi, uint32_t ql[3], uint32_t qh[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double nl = ((double)lo[a] - g.bd[a]) * g.invWd[a], nh = ((double)hi[a] - g.bd[a]) * g.invWd[a];   // exact
        const double fl = floor((nl - 2.0) * 16384.0 - kQnodeMargin), fh = ceil((nh - 2.0) * 16384.0 + kQnodeMargin);
        if (!(fl >= 0.0) || !(fh <= 32767.0) || !(fl <= fh)) return false;
        ql[a] = (uint32_t)fl; qh[a] = (uint32_t)fh;
    }
    return true;
}

__global__ __launch_bounds__(256) void k_write_nodes(const TNode* nodes, uint32_t nn, const uint32_t* new_of_old,
                                                     float4* out_nodes, uint4* out_q, float4* out_n, QGridDev qg,
                                                     double* infl_part, uint32_t* infl_cnt, uint32_t* qfail,
                                                     uint32_t* final_id) {
    __shared__ double s_sum[256];
    __shared__ uint32_t s_cnt[256];
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    double isum = 0.0;
    uint32_t icnt = 0;
    if (g < nn) {
        const TNode& nd = nodes[g];
        const uint32_t i = new_of_old[g];
        float f[16];
        int32_t ref[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int k = 0; k < 6; ++k) f[c * 6 + k] = dec(nd.box[c][k]);
            ref[c] = nd.child[c] >= 0 ? (int32_t)new_of_old[nd.child[c]] : nd.child[c];
            if (ref[c] < 0) {
                // slots of a leaf in triangle-row order (<= 4 entries)
                const uint32_t encl = (uint32_t)(~ref[c]), first = encl >> 3, cnt = encl & 7u;
                for (uint32_t a = 1; a < cnt; ++a) {
                    const uint32_t v = final_id[first + a];
                    uint32_t b = a;
                    while (b > 0 && final_id[first + b - 1] > v) { final_id[first + b] = final_id[first + b - 1]; --b; }
                    final_id[first + b] = v;
                }
            }
        }
        f[12] = __int_as_float(ref[0]); f[13] = __int_as_float(ref[1]); f[14] = 0.f; f[15] = 0.f;
        float4* o = out_nodes + (size_t)i * 4;
        o[0] = make_float4(f[0], f[1], f[2], f[3]); o[1] = make_float4(f[4], f[5], f[6], f[7]);
        o[2] = make_float4(f[8], f[9], f[10], f[11]); o[3] = make_float4(f[12], f[13], f[14], f[15]);
        if (qg.enabled) {
            uint32_t q[8];
            float nf[16];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                uint32_t ql[3], qh[3];
                if (!qbox_dev(qg, f + c * 6, f + c * 6 + 3, ql, qh)) { atomicOr(qfail, 1u); ql[0] = ql[1] = ql[2] = 32767u; qh[0] = qh[1] = qh[2] = 0u; }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    q[c * 4 + a] = ql[a] | (qh[a] << 16);
                    nf[c * 6 + a] = __uint_as_float(0x40000000u | (ql[a] << 8));
                    nf[c * 6 + 3 + a] = __uint_as_float(0x40000000u | (qh[a] << 8));
                }
                q[c * 4 + 3] = (uint32_t)ref[c];
                if (ref[c] < 0) {
                    double hw = 0.0, hq = 0.0;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        hw += (double)f[c * 6 + 3 + a] - (double)f[c * 6 + a];
                        hq += (double)(qh[a] - ql[a]) * qg.cell[a];
                    }
                    if (hw > 0.0) { isum += hq / hw; icnt += 1; }
                }
            }
            nf[12] = f[12]; nf[13] = f[13]; nf[14] = 0.f; nf[15] = 0.f;
            uint4* oq = out_q + (size_t)i * 2;
            oq[0] = make_uint4(q[0], q[1], q[2], q[3]); oq[1] = make_uint4(q[4], q[5], q[6], q[7]);
            float4* on = out_n + (size_t)i * 4;
            on[0] = make_float4(nf[0], nf[1], nf[2], nf[3]); on[1] = make_float4(nf[4], nf[5], nf[6], nf[7]);
            on[2] = make_float4(nf[8], nf[9], nf[10], nf[11]); on[3] = make_float4(nf[12], nf[13], nf[14], nf[15]);
        }
    }
    // mean leaf-box growth: a fixed-shape tree sum (deterministic for a given tree)
    s_sum[threadIdx.x] = isum; s_cnt[threadIdx.x] = icnt;
    __syncthreads();
    for (uint32_t off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) { s_sum[threadIdx.x] += s_sum[threadIdx.x + off]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { infl_part[blockIdx.x] = s_sum[0]; infl_cnt[blockIdx.x] = s_cnt[0]; }
}

// 48-byte triangle records in slot order, ids, labels (same expressions as bvh_build.cpp)
struct QSummary { double isum; uint64_t icnt; uint32_t qfail, pad; };
__global__ __launch_bounds__(256) void k_tri_records(const float* verts3, const uint32_t* tris3, const uint16_t* sem,
                                                     const uint16_t* ins, const uint32_t* final_id, uint32_t T,
                                                     float4* out_tris, uint32_t* slot_prim, uint32_t* slot_label,
                                                     float* slot_box, const double* infl_part, const uint32_t* infl_cnt,
                                                     uint32_t infl_n, const uint32_t* qfail, QSummary* summary) {
    if (blockIdx.x == gridDim.x - 1) {
        // one workgroup more than the records need: the quantisation summary of k_write_nodes (the launch before this one),
        // which the host used to add up after three read-backs.  Thread t adds the partials t, t + 256, ... in that order, a
        // tree over the 256 sums follows: the same value on every run.
        __shared__ double s_sum[256];
        __shared__ unsigned long long s_cnt[256];
        double isum = 0.0;
        unsigned long long icnt = 0;
        for (uint32_t k = threadIdx.x; k < infl_n; k += 256) { isum += infl_part[k]; icnt += infl_cnt[k]; }
        s_sum[threadIdx.x] = isum; s_cnt[threadIdx.x] = icnt;
        __syncthreads();
        for (uint32_t off = 128; off >= 1; off >>= 1) {
            if (threadIdx.x < off) { s_sum[threadIdx.x] += s_sum[threadIdx.x + off]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + off]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { summary->isum = s_sum[0]; summary->icnt = s_cnt[0]; summary->qfail = qfail[0]; summary->pad = 0; }
        return;
    }
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= T + 3u) return;
    float4* o = out_tris + (size_t)s * 3;
    if (s >= T) { o[0] = o[1] = o[2] = make_float4(0.f, 0.f, 0.f, 0.f); return; }     // padding records
    const uint32_t id = final_id[s];
    const float* v0 = verts3 + 3 * (size_t)tris3[3 * (size_t)id];
    const float* v1 = verts3 + 3 * (size_t)tris3[3 * (size_t)id + 1];
    const float* v2 = verts3 + 3 * (size_t)tris3[3 * (size_t)id + 2];
    float a[3], b[3], c[3], e1[3], e2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { a[k] = v0[k]; b[k] = v1[k]; c[k] = v2[k]; e1[k] = a[k] - b[k]; e2[k] = c[k] - a[k]; }
    const float nx = __builtin_fmaf(e2[1], e1[2], -(e2[2] * e1[1]));
    const float ny = __builtin_fmaf(e2[2], e1[0], -(e2[0] * e1[2]));
    const float nz = __builtin_fmaf(e2[0], e1[1], -(e2[1] * e1[0]));
    o[0] = make_float4(a[0], a[1], a[2], e1[0]);
    o[1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
    o[2] = make_float4(e2[2], nx, ny, nz);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        slot_box[(size_t)s * 6 + k] = dec(imin(imin(enc(a[k]), enc(b[k])), enc(c[k])));
        slot_box[(size_t)s * 6 + 3 + k] = dec(imax(imax(enc(a[k]), enc(b[k])), enc(c[k])));
    }
    const uint32_t lab = (sem ? (uint32_t)sem[id] : 0u) | ((ins ? (uint32_t)ins[id] : 0u) << 16);
    slot_prim[s] = id;
    slot_label[s] = lab;
}

// ---- host side -----------------------------------------------------------------------------------------------------------
struct Carver {                      // sub-allocation of the arena, 256-byte aligned
    char* base; size_t off = 0;
    template <class T> T* take(size_t n) {
        T* p = (T*)(base + off);
        off += (n * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

#define DB_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e__ = (call);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            (void)hipGetLastError();                                                                     \
            if (err) *err = std::string("device BVH build: ") + #call + ": " + hipGetErrorString(e__);   \
            return e__ == hipErrorOutOfMemory ? LRC_ERR_OOM : LRC_ERR_HIP;                               \
        }                                                                                                \
    } while (0)

inline int ceil_log2_u64(uint64_t x) { int k = 0; uint64_t p = 1; while (p < x) { p <<= 1; ++k; } return k; }

}  // namespace

void arena_release(DeviceArena* a) {
    if (!a) return;
    if (a->dev) (void)hipFree(a->dev);
    if (a->pinned) (void)hipHostFree(a->pinned);
    if (a->sort_tmp) (void)hipFree(a->sort_tmp);
    *a = DeviceArena();
}

int build_bvh_device(DeviceArena* arena, const float* verts3, uint64_t V, const uint32_t* tris3, uint64_t T64,
                     const uint16_t* tri_sem, const uint16_t* tri_ins, bool on_device, const BuildOptions& opt,
                     int qmode, DeviceScene* out, std::string* err) {
    *out = DeviceScene();
    const int max_leaf = std::min(std::max(opt.max_leaf, 1), kMaxLeaf);
    if (T64 <= (uint64_t)max_leaf || T64 >= (1ull << 28)) return kDevBuildUnsupported;
    const uint32_t T = (uint32_t)T64;
    hipStream_t st = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();

    // ---- arena ----
    const size_t list_cap = (size_t)T / 2 + 2;                  // inner nodes of one level
    const size_t nbig_cap = (size_t)T / kMediumMax + 2;         // big nodes of one level (disjoint ranges > 1024)
    const size_t chunk_cap = (size_t)T / kChunk + nbig_cap + 1;
    size_t need = 0;
    {
        Carver c{nullptr};
        if (!on_device) { c.take<float>(3 * V); c.take<uint32_t>(3 * (size_t)T); c.take<uint16_t>(T); c.take<uint16_t>(T); }
        c.take<PrimRec>(T); c.take<PrimRec>(T); c.take<TNode>(T); c.take<uint32_t>(T);
        for (int k = 0; k < 8; ++k) c.take<uint32_t>(list_cap);
        c.take<BigWork>(nbig_cap); c.take<uint32_t>(nbig_cap + 1); c.take<int>(nbig_cap * 3 * 7 * kBinsN);
        c.take<uint32_t>(nbig_cap);
        c.take<uint32_t>(list_cap / 256 + 2);
        c.take<LevelCounters>(kMaxLevels + 2);
        c.take<uint64_t>(T); c.take<uint64_t>(T); c.take<uint32_t>(T); c.take<uint32_t>(T); c.take<uint32_t>(T);
        c.take<double>((size_t)T / 256 + 2); c.take<uint32_t>((size_t)T / 256 + 2); c.take<uint32_t>(4);
        c.take<SNode>(T); c.take<SubRoot>(list_cap); c.take<SubInfo>(list_cap); c.take<uint32_t>(list_cap); c.take<uint32_t>(16);
        need = c.off;
    }
    if (arena->cap < need) {
        if (arena->dev) { (void)hipFree(arena->dev); arena->dev = nullptr; arena->cap = 0; }
        DB_HIP(hipMalloc(&arena->dev, need + need / 8));
        arena->cap = need + need / 8;
    }
    if (!arena->pinned) DB_HIP(hipHostMalloc(&arena->pinned, 4096, hipHostMallocDefault));
    Carver c{(char*)arena->dev};
    const float* d_verts = verts3;
    const uint32_t* d_tris = tris3;
    const uint16_t* d_sem = tri_sem;
    const uint16_t* d_ins = tri_ins;
    if (!on_device) {
        float* dv = c.take<float>(3 * V);
        uint32_t* dt = c.take<uint32_t>(3 * (size_t)T);
        uint16_t* ds = c.take<uint16_t>(T);
        uint16_t* di = c.take<uint16_t>(T);
        DB_HIP(hipMemcpyAsync(dv, verts3, 3 * V * 4, hipMemcpyHostToDevice, st));
        DB_HIP(hipMemcpyAsync(dt, tris3, 3 * (size_t)T * 4, hipMemcpyHostToDevice, st));
        if (tri_sem) DB_HIP(hipMemcpyAsync(ds, tri_sem, (size_t)T * 2, hipMemcpyHostToDevice, st));
        if (tri_ins) DB_HIP(hipMemcpyAsync(di, tri_ins, (size_t)T * 2, hipMemcpyHostToDevice, st));
        d_verts = dv; d_tris = dt; d_sem = tri_sem ? ds : nullptr; d_ins = tri_ins ? di : nullptr;
    }
    PrimRec* bufA = c.take<PrimRec>(T);
    PrimRec* bufB = c.take<PrimRec>(T);
    TNode* nodes = c.take<TNode>(T);
    uint32_t* final_id = c.take<uint32_t>(T);
    uint32_t* lists[2][4];                  // per level parity: small, medium, big, tiny
    for (int a = 0; a < 2; ++a) for (int k = 0; k < 4; ++k) lists[a][k] = c.take<uint32_t>(list_cap);
    BigWork* work = c.take<BigWork>(nbig_cap);
    uint32_t* chunk_start = c.take<uint32_t>(nbig_cap + 1);
    int* gbins = c.take<int>(nbig_cap * 3 * 7 * kBinsN);
    uint32_t* median_list = c.take<uint32_t>(nbig_cap);
    uint32_t* partial = c.take<uint32_t>(list_cap / 256 + 2);
    // [0]: the scene bounds (8 ints), [1 ...]: the level counters -- bounds and level 0 come back in one copy
    LevelCounters* counters = c.take<LevelCounters>(kMaxLevels + 2) + 1;
    int* bounds = (int*)(counters - 1);
    static_assert(sizeof(LevelCounters) >= 32, "the bounds share a LevelCounters slot");
    uint64_t* keys_in = c.take<uint64_t>(T);
    uint64_t* keys_out = c.take<uint64_t>(T);
    uint32_t* vals_in = c.take<uint32_t>(T);
    uint32_t* vals_out = c.take<uint32_t>(T);
    uint32_t* new_of_old = c.take<uint32_t>(T);
    double* infl_part = c.take<double>((size_t)T / 256 + 2);
    uint32_t* infl_cnt = c.take<uint32_t>((size_t)T / 256 + 2);
    uint32_t* qfail = c.take<uint32_t>(4);
    SNode* sub_stage = c.take<SNode>(T);
    SubRoot* sub_roots = c.take<SubRoot>(list_cap);
    SubInfo* sub_info = c.take<SubInfo>(list_cap);
    uint32_t* sub_bases = c.take<uint32_t>(list_cap);
    uint32_t* sub_count = c.take<uint32_t>(16);            // [0] roots so far; [4..7] SubTotals; [8..13] QSummary
    {
        size_t tmp = 0;
        (void)rocprim::radix_sort_pairs(nullptr, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)T, 0, 64, st);
        if (arena->sort_cap < tmp) {
            if (arena->sort_tmp) { (void)hipFree(arena->sort_tmp); arena->sort_tmp = nullptr; arena->sort_cap = 0; }
            DB_HIP(hipMalloc(&arena->sort_tmp, tmp + 256));
            arena->sort_cap = tmp + 256;
        }
    }

    // ---- level 0 ----
    {
        SetupArgs sa{};
        sa.nodes = nodes; sa.counters = counters; sa.bounds = bounds; sa.qfail = qfail; sa.sub_count = sub_count; sa.T = T;
        for (int k = 0; k < 4; ++k) sa.list0[k] = lists[0][k];
        hipLaunchKernelGGL(k_setup, dim3(1), dim3(256), 0, st, sa);
    }
    hipLaunchKernelGGL(k_check_verts, dim3((uint32_t)((3 * V + 255) / 256)), dim3(256), 0, st, d_verts, 3 * V, counters);
    hipLaunchKernelGGL(k_init_prims, dim3((T + 255) / 256), dim3(256), 0, st, d_verts, V, d_tris, T, bufA, nodes, bounds,
                       counters);
    struct Landing { LevelCounters b_lc[2]; LevelCounters lc; LevelCounters lc_pair[2]; int bounds[8]; uint32_t median[64]; };
    Landing* land = (Landing*)arena->pinned;
    DB_HIP(hipMemcpyAsync(land->b_lc, counters - 1, 2 * sizeof(LevelCounters), hipMemcpyDeviceToHost, st));
    DB_HIP(hipStreamSynchronize(st));
    std::memcpy(land->bounds, &land->b_lc[0], 32);
    land->lc = land->b_lc[1];
    if (land->lc.error & 1u) { if (err) *err = "lrc_scene_create: triangle index out of range"; return LRC_ERR_INVALID_ARG; }
    if (land->lc.error & 2u) { if (err) *err = "lrc_scene_create: vertex coordinate is not finite or exceeds 1e6"; return LRC_ERR_INVALID_ARG; }
    auto host_dec = [](int i) { int u = i ^ ((i >> 31) & 0x7FFFFFFF); float f; std::memcpy(&f, &u, 4); return f; };
    for (int k = 0; k < 3; ++k) { out->bounds_lo[k] = host_dec(land->bounds[k]); out->bounds_hi[k] = host_dec(land->bounds[3 + k]); }
    const auto t_uploaded = std::chrono::steady_clock::now();

    Params P{};
    P.nodes = nodes; P.final_id = final_id; P.max_leaf = max_leaf; P.median_only = opt.median_only;
    {
        const uint64_t leaves = ((uint64_t)T + max_leaf - 1) / max_leaf;
        const int balanced = ceil_log2_u64(leaves ? leaves : 1);
        const int cap = opt.depth_slack >= 0 ? balanced + opt.depth_slack : kMaxDepth - 1;
        P.depth_cap = std::min(std::max(cap, balanced), kMaxDepth - 1);
    }
    PrimRec* cur = bufA;
    PrimRec* nxt = bufB;
    const bool sub_enabled = opt.subtrees != 0;
    uint32_t n_sub_roots = 0;
    uint32_t base = 0, level = 0;
    uint64_t num_leaves = 0;
    uint32_t max_leaf_seen = 0;
    LevelCounters lc = land->lc;
    bool bins_clean = false;
    while (lc.n_nodes > 0) {
        if (level + 1 >= (uint32_t)kMaxLevels || (uint64_t)base + lc.n_nodes > T) {
            if (err) *err = "device BVH build: level bookkeeping out of range";
            return LRC_ERR_INTERNAL;
        }
        P.cur = cur; P.nxt = nxt;
        uint32_t** L = lists[level & 1];
        uint32_t** Ln = lists[(level + 1) & 1];
        LevelCounters* cl = counters + level;
        // ONE host synchronisation per level: the counters of this level (has the SAH failed on a big node?) and of the next
        // come back together after the whole level has been enqueued.  A big node the SAH could not split (never on the
        // meshes measured; `median_only` forces it) is ranked by a radix sort of its segment afterwards, and the level's
        // emission -- which copied that node's children's bounds -- is done again.
        const uint32_t nb = lc.n_big;
        const uint32_t max_chunks = T / kChunk + nb + 1;
        if (nb) {
            hipLaunchKernelGGL(k_big_prefix, dim3(1), dim3(1024), 0, st, nodes, L[2], nb, work, chunk_start);
            if (!bins_clean) {
                const uint32_t total = nb * 3 * 7 * kBinsN;
                hipLaunchKernelGGL(k_big_bins_init, dim3((total + 255) / 256), dim3(256), 0, st, gbins, total);
            }
            hipLaunchKernelGGL(k_big_bin, dim3(max_chunks), dim3(256), 0, st, P, work, chunk_start, nb, gbins);
            hipLaunchKernelGGL(k_big_eval, dim3(nb), dim3(192), 0, st, P, work, gbins, cl, median_list, (uint32_t)nbig_cap);
            hipLaunchKernelGGL(k_big_scatter, dim3(max_chunks), dim3(256), 0, st, P, work, chunk_start, nb);
            bins_clean = true;           // slots [0, 2 nb) are empty again
        }
        if (lc.n_medium) hipLaunchKernelGGL(k_medium, dim3(lc.n_medium), dim3(256), 0, st, P, (const uint32_t*)L[1]);
        if (lc.n_small)
            hipLaunchKernelGGL(k_small, dim3((lc.n_small + 3) / 4), dim3(256), 0, st, P, (const uint32_t*)L[0], lc.n_small);
        if (lc.n_tiny)
            hipLaunchKernelGGL(k_tiny, dim3((lc.n_tiny + 15) / 16), dim3(256), 0, st, P, (const uint32_t*)L[3], lc.n_tiny);
        const uint32_t nblk = (lc.n_nodes + 255) / 256;
        // Children of <= 64 primitives become subtree roots (k_subtree, one launch after the last level) once the numbers
        // handed out lie beyond the breadth-first head of the final layout, whose numbers must stay breadth-first.
        const int sub_mode = sub_enabled && (uint64_t)base + lc.n_nodes > (uint64_t)std::max(opt.bfs_nodes, 1) ? 1 : 0;
        auto emit = [&]() {
            // a level of at most 256 nodes is one workgroup: nothing in front of it to count
            if (nblk > 1)
                hipLaunchKernelGGL(k_emit_count, dim3(nblk), dim3(256), 0, st, (const TNode*)nodes, base, lc.n_nodes, max_leaf, partial);
            hipLaunchKernelGGL(k_emit_write, dim3(nblk), dim3(256), 0, st, nodes, base, lc.n_nodes, base + lc.n_nodes, max_leaf,
                               (const uint32_t*)partial, (const BigWork*)work, Ln[3], Ln[0], Ln[1], Ln[2], counters + level + 1,
                               sub_mode, sub_roots, sub_count, (level + 1) & 1u);
        };
        emit();
        DB_HIP(hipMemcpyAsync(land->lc_pair, cl, 2 * sizeof(LevelCounters), hipMemcpyDeviceToHost, st));
        DB_HIP(hipStreamSynchronize(st));
        if (nb && land->lc_pair[0].n_median) {
            const uint32_t nmed = land->lc_pair[0].n_median;
            std::vector<uint32_t> slots(nmed);
            std::vector<BigWork> hw(nb);
            DB_HIP(hipMemcpy(slots.data(), median_list, nmed * 4, hipMemcpyDeviceToHost));
            DB_HIP(hipMemcpy(hw.data(), work, nb * sizeof(BigWork), hipMemcpyDeviceToHost));
            for (uint32_t q = 0; q < nmed; ++q) {
                const BigWork& w = hw[slots[q]];
                TNode hn;
                DB_HIP(hipMemcpy(&hn, nodes + w.g, sizeof(TNode), hipMemcpyDeviceToHost));
                const uint32_t n = hn.end - hn.begin;
                hipLaunchKernelGGL(k_median_keys, dim3((n + 255) / 256), dim3(256), 0, st, (const PrimRec*)cur, hn.begin,
                                   n, w.axis, keys_in, vals_in);
                size_t tmp = arena->sort_cap;
                hipError_t se = rocprim::radix_sort_pairs(arena->sort_tmp, tmp, keys_in, keys_out, vals_in, vals_out,
                                                          (size_t)n, 0, 64, st);
                DB_HIP(se);
                hipLaunchKernelGGL(k_median_gather, dim3((n + 255) / 256), dim3(256), 0, st, P, work + slots[q], n,
                                   (const uint32_t*)vals_out);
            }
            // the emission again, from the state before it: the next level's counters and the running count of subtree roots
            DB_HIP(hipMemsetAsync(counters + level + 1, 0, sizeof(LevelCounters), st));
            DB_HIP(hipMemcpyAsync(sub_count, &n_sub_roots, 4, hipMemcpyHostToDevice, st));
            emit();
            DB_HIP(hipMemcpyAsync(land->lc_pair, cl, 2 * sizeof(LevelCounters), hipMemcpyDeviceToHost, st));
            DB_HIP(hipStreamSynchronize(st));
        }
        land->lc = land->lc_pair[1];
        base += lc.n_nodes;
        lc = land->lc;
        num_leaves += lc.n_leaves;
        max_leaf_seen = std::max(max_leaf_seen, lc.max_leaf);
        if (lc.n_tiny + lc.n_small + lc.n_medium + lc.n_big + lc.n_sub != lc.n_nodes) {
            if (err) *err = "device BVH build: work lists do not add up";
            return LRC_ERR_INTERNAL;
        }
        n_sub_roots += lc.n_sub;
        std::swap(cur, nxt);
        ++level;
        if (lc.n_nodes > 0 && lc.n_sub == lc.n_nodes) {        // a level of subtree roots only: nothing left for the grid
            base += lc.n_nodes;
            ++level;
            break;
        }
    }
    DB_HIP(hipGetLastError());
    uint32_t nn = base;
    if (n_sub_roots) {
        // everything below the subtree roots in one launch; then the staged nodes get their numbers behind the level-numbered ones
        hipLaunchKernelGGL(k_subtree, dim3((n_sub_roots + 3) / 4), dim3(256), 0, st, P, (const PrimRec*)bufA, (const PrimRec*)bufB,
                           (const SubRoot*)sub_roots, n_sub_roots, sub_stage, sub_info);
        hipLaunchKernelGGL(k_sub_scan, dim3(1), dim3(1024), 0, st, (const SubInfo*)sub_info, n_sub_roots, sub_bases,
                           (SubTotals*)(sub_count + 4));
        hipLaunchKernelGGL(k_sub_place, dim3((n_sub_roots + 3) / 4), dim3(256), 0, st, nodes, (const SubRoot*)sub_roots, n_sub_roots,
                           (const SNode*)sub_stage, (const SubInfo*)sub_info, (const uint32_t*)sub_bases, nn);
        DB_HIP(hipMemcpyAsync(land->median, sub_count + 4, sizeof(SubTotals), hipMemcpyDeviceToHost, st));
        DB_HIP(hipStreamSynchronize(st));
        SubTotals tot;
        std::memcpy(&tot, land->median, sizeof(tot));
        if ((uint64_t)nn + tot.nodes > T) {
            if (err) *err = "device BVH build: subtree bookkeeping out of range";
            return LRC_ERR_INTERNAL;
        }
        nn += tot.nodes;
        num_leaves += tot.leaves;
        max_leaf_seen = std::max(max_leaf_seen, tot.max_leaf);
        level = std::max(level, tot.max_depth);
    }
    const auto t_tree = std::chrono::steady_clock::now();

    // ---- the scene arrays ----
    QGridDev qg{};
    QGrid hg;
    qg.enabled = 0;
    if (qmode != 0 && make_qgrid_bounds(out->bounds_lo, out->bounds_hi, out->qbase, out->qW, out->qinvW, hg)) {
        qg.enabled = 1;
        for (int a = 0; a < 3; ++a) { qg.bd[a] = hg.bd[a]; qg.invWd[a] = 1.0 / hg.Wd[a]; qg.cell[a] = hg.Wd[a] / 16384.0; }
    }
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_nodes = al((size_t)nn * 64), b_q = qg.enabled ? al((size_t)nn * 32) : 0, b_n = qg.enabled ? al((size_t)nn * 64) : 0;
    const size_t b_tris = al(((size_t)T + 3) * 48), b_id = al((size_t)T * 4);
    const size_t b_box = al((size_t)T * 24);
    const size_t slab_bytes = b_nodes + b_q + b_n + b_tris + 2 * b_id + b_box;
    void* slab = nullptr;
    DB_HIP(hipMalloc(&slab, slab_bytes));
    char* sp = (char*)slab;
    out->slab = slab; out->slab_bytes = slab_bytes;
    out->nodes = sp; sp += b_nodes;
    out->nodes_q = qg.enabled ? sp : nullptr; sp += b_q;
    out->nodes_n = qg.enabled ? sp : nullptr; sp += b_n;
    out->tris = sp; sp += b_tris;
    out->slot_prim = (uint32_t*)sp; sp += b_id;
    out->slot_label = (uint32_t*)sp; sp += b_id;
    out->slot_box = b_box ? (float*)sp : nullptr;
    auto bail = [&](int rc) { (void)hipFree(slab); *out = DeviceScene(); return rc; };

    const uint32_t head = std::min<uint32_t>((uint32_t)std::max(opt.bfs_nodes, 1), nn);
    if (head < nn) {
        const uint32_t nt = nn - head;
        hipLaunchKernelGGL(k_tail_keys, dim3((nt + 255) / 256), dim3(256), 0, st, (const TNode*)nodes, head, nn, keys_in, vals_in);
        size_t tmp = arena->sort_cap;
        hipError_t se = rocprim::radix_sort_pairs(arena->sort_tmp, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)nt, 0, 41, st);
        if (se != hipSuccess) { if (err) *err = std::string("device BVH build: radix sort: ") + hipGetErrorString(se); return bail(LRC_ERR_HIP); }
    }
    const uint32_t nblk = (nn + 255) / 256;
    hipLaunchKernelGGL(k_new_index, dim3(nblk), dim3(256), 0, st, head, nn, (const uint32_t*)vals_out, new_of_old);
    hipLaunchKernelGGL(k_write_nodes, dim3(nblk), dim3(256), 0, st, (const TNode*)nodes, nn, (const uint32_t*)new_of_old,
                       (float4*)out->nodes, (uint4*)out->nodes_q, (float4*)out->nodes_n, qg, infl_part, infl_cnt, qfail, final_id);
    hipLaunchKernelGGL(k_tri_records, dim3((T + 3 + 255) / 256 + 1), dim3(256), 0, st, d_verts, d_tris, d_sem, d_ins,
                       (const uint32_t*)final_id, T, (float4*)out->tris, out->slot_prim, out->slot_label, out->slot_box,
                       (const double*)infl_part, (const uint32_t*)infl_cnt, nblk, (const uint32_t*)qfail, (QSummary*)(sub_count + 8));
    const hipError_t e_copy = hipMemcpyAsync(land->median, sub_count + 8, sizeof(QSummary), hipMemcpyDeviceToHost, st);
    const hipError_t e_sync = hipStreamSynchronize(st);
    {
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = e_copy != hipSuccess ? e_copy : e_sync;
        if (e != hipSuccess) { (void)hipGetLastError(); if (err) *err = std::string("device BVH build: ") + hipGetErrorString(e); return bail(LRC_ERR_HIP); }
    }
    QSummary qs;
    std::memcpy(&qs, land->median, sizeof(qs));
    const double isum = qs.isum;
    const uint64_t icnt = qs.icnt;
    const uint32_t hq = qs.qfail;
    out->leaf_inflation = icnt ? isum / (double)icnt : 1.0;
    if (qg.enabled && (hq != 0 || (qmode < 2 && out->leaf_inflation > kQnodeMaxInflation))) {
        out->nodes_q = nullptr; out->nodes_n = nullptr;      // the float32 nodes serve alone (the bytes stay in the slab)
    }
    out->num_nodes = nn;
    out->num_leaves = num_leaves;
    out->num_slots = T;
    out->max_depth = level;            // inner nodes down to depth level - 1, their leaves one below
    out->max_leaf_size = max_leaf_seen;
    out->levels = level;
    const auto t_end = std::chrono::steady_clock::now();
    out->ms_upload = std::chrono::duration<double, std::milli>(t_uploaded - t_begin).count();
    out->ms_hierarchy = std::chrono::duration<double, std::milli>(t_tree - t_uploaded).count();
    out->ms_emit = std::chrono::duration<double, std::milli>(t_end - t_tree).count();
    return kDevBuildOk;
}

}  // namespace lrc

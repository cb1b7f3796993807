// lrc_bvh.h -- host-side BVH builder of liblidarcast (product code, not the oracle).
//
// Stands in for what Embree does inside open3d RaycastingScene.add_triangles
// (reference call site raycast_engine/raycast_engine_cpu.py:46-47), but runs ONCE per mesh.
//
// Layout produced:
//   nodes   : num_nodes x 16 float   (64 B) binary BVH, both child boxes in the parent
//             [0..2] lo0  [3..5] hi0  [6..8] lo1  [9..11] hi1  [12] ref0  [13] ref1  [14..15] 0
//             ref (int32 bits): >= 0 inner node index; < 0 leaf, ~ref = first_slot*8 + count
//   tri_rec : num_slots x 12 float   (48 B) per leaf slot: v0, v1, v2, Ng = cross(v2-v0, v0-v1)
//   slot_prim / slot_label : per slot, the caller's triangle row and (sem | ins<<16)
#pragma once
#include <cstdint>
#include <vector>

namespace lrc {

constexpr int kMaxDepth = 32;        // == LRC_MAX_BVH_DEPTH: leaf depth <= 31, stack of 32 suffices
constexpr int kMaxLeaf  = 4;
constexpr int kNodeFloats = 16;
constexpr int kTriFloats  = 12;

struct HostBVH {
    std::vector<float>    nodes;       // 16 floats per inner node, node 0 = root
    std::vector<float>    tri_rec;     // 12 floats per slot
    std::vector<uint32_t> slot_prim;
    std::vector<uint32_t> slot_label;
    std::vector<float>    slot_box;    // 6 floats per slot, the triangle's exact vertex box (lo xyz, hi xyz)
    uint64_t num_nodes = 0, num_leaves = 0, num_slots = 0;
    uint32_t max_depth = 0, max_leaf_size = 0;
    float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};
};

struct BuildOptions {
    int max_leaf  = kMaxLeaf;   // 1..4
    int bfs_nodes = 2048;       // nodes laid out breadth-first at the front (the part all rays share)
    int threads   = 0;          // 0 = min(hardware threads, 16)
    int depth_slack = 2;        // >= 0: leaf depth <= balanced-tree height + slack; -1: only the hard cap (31)
    int median_only = 0;        // test hook: every split is the median fallback (exercises that path in both builders)
    int subtrees = 1;           // device builder: nodes of <= 64 primitives beyond the breadth-first head are finished by one
                                // wave each in ONE launch (k_subtree) instead of level by level; 0 = level by level throughout
};

// verts3: V x 3 float32, tris3: T x 3 uint32 (validated by the caller).  Deterministic.
void build_bvh(const float* verts3, uint64_t V, const uint32_t* tris3, uint64_t T,
               const uint16_t* tri_sem, const uint16_t* tri_ins,
               const BuildOptions& opt, HostBVH* out);

}  // namespace lrc

// lrc_bvh_device.h -- the scene build on the GPU (product code).
//
// Stands in for what Embree does inside open3d RaycastingScene.add_triangles (reference call site
// raycast_engine/raycast_engine_cpu.py:46-47).  Same algorithm as the host builder (bvh_build.cpp: binned SAH, 64 bins
// x 3 axes, median fallback, leaves <= max_leaf, depth cap), evaluated level by level on the device; every decision
// is a function of order-independent quantities (min / max of boxes, integer counts, double-precision costs swept in
// bin order), so the tree -- and with the canonical layout both builders share, every byte of the node, triangle and
// id arrays -- equals the host builder's (tests/test_parity_gpu.py::test_device_build_equals_host_build).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "lrc_bvh.h"

namespace lrc {

// scratch of the device builder, owned by the context and reused from scene to scene
struct DeviceArena {
    void* dev = nullptr;
    size_t cap = 0;
    void* pinned = nullptr;        // small page-locked landing area of the per-level counters
    void* sort_tmp = nullptr;      // rocPRIM temporary storage
    size_t sort_cap = 0;
};
void arena_release(DeviceArena* a);

struct DeviceScene {               // one allocation (`slab`), the arrays point into it
    void* slab = nullptr;
    size_t slab_bytes = 0;
    void* nodes = nullptr;         // num_nodes x 64 B
    void* tris = nullptr;          // (num_slots + 3) x 48 B
    uint32_t* slot_prim = nullptr;
    uint32_t* slot_label = nullptr;
    float* slot_box = nullptr;     // num_slots x 24 B
    void* nodes_q = nullptr;       // num_nodes x 32 B, or NULL
    void* nodes_n = nullptr;       // num_nodes x 64 B, or NULL
    uint64_t num_nodes = 0, num_leaves = 0, num_slots = 0;
    uint32_t max_depth = 0, max_leaf_size = 0, levels = 0;
    float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};
    float qbase[3] = {0, 0, 0}, qW[3] = {1, 1, 1}, qinvW[3] = {1, 1, 1};
    double leaf_inflation = 1.0;
    double ms_upload = 0, ms_hierarchy = 0, ms_emit = 0;
};

enum { kDevBuildOk = 0, kDevBuildUnsupported = 1 };    // negative: an LRC_ERR_* code, text in *err

// verts3 / tris3 / labels: HOST pointers when on_device is false (they are uploaded), device pointers otherwise.
// qmode: 0 = float32 nodes only, 1 = quantised images when the grid is fine enough, 2 = whenever the grid fits.
// Synchronous: on return the scene arrays are complete.  kDevBuildUnsupported: the caller uses the host builder
// (a mesh of <= max_leaf triangles).
int build_bvh_device(DeviceArena* arena, const float* verts3, uint64_t V, const uint32_t* tris3, uint64_t T,
                     const uint16_t* tri_sem, const uint16_t* tri_ins, bool on_device, const BuildOptions& opt,
                     int qmode, DeviceScene* out, std::string* err);

}  // namespace lrc

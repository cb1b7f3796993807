// lrc_qnodes.h -- quantised node images of the host BVH (product code; DESIGN.md section 4.1, "32-byte quantised node images").
// Host only, no HIP: built into liblidarcast and, for the CPU tests, into tests/native/sanitize_harness.cpp.
#pragma once
#include <cstdint>
#include <vector>

#include "lrc_bvh.h"

namespace lrc {

constexpr double kQnodeMaxInflation = 1.05;   // images are built while the mean growth of a leaf box stays below this
constexpr double kQnodeMargin = 1.0 / 16;     // cells; see the bound in DESIGN.md section 4.1
// The bound the margin is proven for, in units of W per axis: a ray takes the quantised path only if its origin o has
// |o - base| <= kQnodeNearBase * W and |o| <= kQnodeNearOrigin * W (and |d| <= 2^60); the builder refuses a scene whose
// base is farther than kQnodeMaxBase * W from the world origin (an origin inside such a scene would fail the second test).
// 2^-24 (4 * 6 + 12 + 8) W = 2.7e-6 W < margin * cell = 2^-18 W.
constexpr float kQnodeNearBase = 6.0f;
constexpr float kQnodeNearOrigin = 12.0f;
constexpr double kQnodeMaxBase = 8.0;
static_assert((4.0 * kQnodeNearBase + kQnodeNearOrigin + 8.0) / 16777216.0 < kQnodeMargin / 16384.0,
              "the margin must cover the arithmetic bound");
static_assert(kQnodeMaxBase + 4.0 <= kQnodeNearOrigin, "origins inside the scene must pass the |o| test");

struct QGrid { double Wd[3], bd[3]; };        // per axis: cell width * 2^14 (a power of two) and the float32 base

// the grid of a tree: base / W / 1/W as the kernel takes them.  false: the scene does not fit a grid
bool make_qgrid(const HostBVH& h, float base[3], float W[3], float invW[3], QGrid& g);
// the same from the scene bounds alone (the device builder has no HostBVH)
bool make_qgrid_bounds(const float lo[3], const float hi[3], float base[3], float W[3], float invW[3], QGrid& g);

// 32-byte nodes (8 words per node) and the same boxes as normalised float32 (16 floats per node)
bool build_qnodes(const HostBVH& h, const QGrid& g, std::vector<uint32_t>& q8, std::vector<float>& n16,
                  double* leaf_inflation);

// the four-wide collapse: 64-byte nodes (16 words) and 128-byte float32 nodes (32 floats)
bool build_q4nodes(const HostBVH& h, const QGrid& g, std::vector<uint32_t>& q16, std::vector<float>& n32, uint64_t* num4);

}  // namespace lrc

// lrc_qnodes.h -- quantised node images of the host BVH (product code; DESIGN.md section 4.1, "32-byte quantised node images").
// Host only, no HIP: built into liblidarcast and, for the CPU tests, into tests/native/sanitize_harness.cpp.
#pragma once
#include <cstdint>
#include <vector>

#include "lrc_bvh.h"

namespace lrc {

constexpr double kQnodeMaxInflation = 1.05;   // images are built while the mean growth of a leaf box stays below this
constexpr double kQnodeMargin = 1.0 / 16;     // cells; see the bound in DESIGN.md section 4.1

struct QGrid { double Wd[3], bd[3]; };        // per axis: cell width * 2^14 (a power of two) and the float32 base

// the grid of a tree: base / W / 1/W as the kernel takes them.  false: the scene does not fit a grid
bool make_qgrid(const HostBVH& h, float base[3], float W[3], float invW[3], QGrid& g);

// 32-byte nodes (8 words per node) and the same boxes as normalised float32 (16 floats per node)
bool build_qnodes(const HostBVH& h, const QGrid& g, std::vector<uint32_t>& q8, std::vector<float>& n16,
                  double* leaf_inflation);

// the four-wide collapse: 64-byte nodes (16 words) and 128-byte float32 nodes (32 floats)
bool build_q4nodes(const HostBVH& h, const QGrid& g, std::vector<uint32_t>& q16, std::vector<float>& n32, uint64_t* num4);

}  // namespace lrc

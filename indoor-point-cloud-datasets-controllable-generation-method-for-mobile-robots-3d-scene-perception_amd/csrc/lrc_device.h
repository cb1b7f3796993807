// lrc_device.h -- device-side arithmetic of the ray-cast scan (gfx950 only).
//
// This header fixes the float32 expression trees that define a "hit" (DESIGN.md section 3).
// oracle/lrc_oracle.c restates the same trees in plain C for the CPU; tests compare bit for bit.
// The file is compiled with -ffp-contract=off: every fused multiply-add below is explicit.
//
// Reference semantics being reproduced:
//   * closest hit, two-sided, tnear = 0 exclusive, tfar = +inf, parametric t along the given
//     direction: the contract of open3d RaycastingScene.cast_rays, called at
//     raycast_engine/raycast_engine_cpu.py:51 (Embree's Moeller-Trumbore intersector restated).
//   * p = o + (d/|d|)*t in float32, separate mul and add:   raycast_engine_cpu.py:55-62
//   * range filter and incident angle in float64:            raycast_engine_cpu.py:95-107
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lrcdev {

#define LRC_DI __device__ __forceinline__

constexpr float kTinyDir = 1e-30f;
constexpr float kPadRelLo = 0.999755859375f;     // 1 - 2^-12
constexpr float kPadRelHi = 1.000244140625f;     // 1 + 2^-12
constexpr float kPadAbs   = 1.52587890625e-05f;  // 2^-16
constexpr double kRadToDeg = 57.29577951308232;  // 180/pi, numpy's npy_rad2deg factor

struct V3 { float x, y, z; };

LRC_DI float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
LRC_DI float dot3(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
LRC_DI V3 cross3(V3 a, V3 b) {
    V3 r;
    r.x = fma_(a.y, b.z, -(a.z * b.y));
    r.y = fma_(a.z, b.x, -(a.x * b.z));
    r.z = fma_(a.x, b.y, -(a.y * b.x));
    return r;
}
LRC_DI V3 sub3(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
LRC_DI float xorsign(float x, uint32_t s) { return __uint_as_float(__float_as_uint(x) ^ s); }
LRC_DI float min2(float a, float b) { return a < b ? a : b; }
LRC_DI float max2(float a, float b) { return a > b ? a : b; }

// Per-ray constants of the slab test.
struct RaySlab { float ix, iy, iz, ox, oy, oz; };   // idir and o*idir

LRC_DI float safe_inv(float d) {
    float a = __builtin_fabsf(d);
    float s = a < kTinyDir ? __builtin_copysignf(kTinyDir, d) : d;
    return 1.0f / s;
}
LRC_DI RaySlab make_slab(V3 o, V3 d) {
    RaySlab r;
    r.ix = safe_inv(d.x); r.iy = safe_inv(d.y); r.iz = safe_inv(d.z);
    r.ox = o.x * r.ix; r.oy = o.y * r.iy; r.oz = o.z * r.iz;
    return r;
}

// which of lo / hi a ray enters a box through, per axis: bit a set = negative 1/d (safe_inv keeps the sign of d, of a
// zero component too)
LRC_DI uint32_t sign_octant(V3 d) {
    return (__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2);
}

// Padded entry/exit parameters of the ray against the box [lo,hi].
// Monotone in lo/hi, hence the interval of a box contains the interval of every box inside it.
LRC_DI void slab_interval(const RaySlab& s, float lox, float loy, float loz, float hix, float hiy,
                          float hiz, float& tn, float& tf) {
    float t0x = fma_(lox, s.ix, -s.ox), t1x = fma_(hix, s.ix, -s.ox);
    float t0y = fma_(loy, s.iy, -s.oy), t1y = fma_(hiy, s.iy, -s.oy);
    float t0z = fma_(loz, s.iz, -s.oz), t1z = fma_(hiz, s.iz, -s.oz);
    float nx = min2(t0x, t1x), fx = max2(t0x, t1x);
    float ny = min2(t0y, t1y), fy = max2(t0y, t1y);
    float nz = min2(t0z, t1z), fz = max2(t0z, t1z);
    float n = max2(max2(nx, ny), max2(nz, 0.0f));
    float f = min2(min2(fx, fy), fz);
    tn = fma_(n, kPadRelLo, -kPadAbs);
    tf = fma_(f, kPadRelHi, kPadAbs);
}

// Ray/triangle test on an edge record (v0, e1 = v0 - v1, e2 = v2 - v0, Ng = cross(e2, e1)); the hit definition of
// DESIGN.md section 3 is tri_mt && box_clause:
//   den = Ng.D != 0,  U = (C x D).e2,  V = (C x D).e1 (sign-corrected), U,V >= 0, U+V <= |den|,
//   T = Ng.C (sign-corrected) > 0,  t = T/|den| finite  [tri_mt: the Moeller-Trumbore conditions, Embree's form]
//   and t inside the padded slab interval of the triangle's own vertex box  [box_clause: the one clause Embree lacks]
// The traversal ranks candidates by tri_mt alone and tests the clause ONCE, on the closest candidate, after the
// traversal: if it passes, that candidate is the definition's closest hit (every triangle passing both tests is a
// candidate too, and pruning by a candidate's t never hides a closer one); if it fails -- never observed -- the ray is
// redone with the clause tested per triangle.  The edges are stored, not formed per test: six subtractions and seven
// registers less in the hot loop; the vertex box the clause needs sits in a side table (slot_box).
LRC_DI bool tri_mt(V3 o, V3 d, V3 v0, V3 e1, V3 e2, V3 ng, float& t_out) {
    V3 c = sub3(v0, o);
    V3 r = cross3(c, d);
    float den = dot3(ng, d);
    float aden = __builtin_fabsf(den);
    uint32_t sgn = __float_as_uint(den) & 0x80000000u;
    float u = xorsign(dot3(r, e2), sgn);
    float v = xorsign(dot3(r, e1), sgn);
    float tt = xorsign(dot3(ng, c), sgn);
    bool ok = (den != 0.0f) & (u >= 0.0f) & (v >= 0.0f) & (u + v <= aden) & (tt > 0.0f);
    if (!ok) return false;
    float t = tt / aden;
    if (!(t < __builtin_inff())) return false;
    t_out = t;
    return true;
}
LRC_DI bool box_clause(const RaySlab& s, float lox, float loy, float loz, float hix, float hiy, float hiz, float t) {
    float tn, tf;
    slab_interval(s, lox, loy, loz, hix, hiy, hiz, tn, tf);
    return (tn <= t) & (t <= tf);
}
// both at once, for callers that test few (ray, triangle) pairs (laboratory kernels): bx = the slot's box, 6 floats
LRC_DI bool tri_hit(V3 o, V3 d, const RaySlab& s, V3 v0, V3 e1, V3 e2, V3 ng, const float* bx, float& t_out) {
    float t;
    if (!tri_mt(o, d, v0, e1, e2, ng, t)) return false;
    if (!box_clause(s, bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], t)) return false;
    t_out = t;
    return true;
}

// A ray takes part in the cast only if its six components are finite (bit test: immune to -fno-honor-nans).  Anything
// else is reported as a miss without touching the tree, so no comparison ever sees a NaN (include/lidarcast.h,
// "finite-ray contract"; the oracle applies the same rule).
LRC_DI bool finite_ray(V3 o, V3 d) {
    const uint32_t m = 0x7F800000u;
    return ((__float_as_uint(o.x) & m) != m) & ((__float_as_uint(o.y) & m) != m) & ((__float_as_uint(o.z) & m) != m) &
           ((__float_as_uint(d.x) & m) != m) & ((__float_as_uint(d.y) & m) != m) & ((__float_as_uint(d.z) & m) != m);
}

// One output of np.dot(directions, R.T): the BLAS kernel accumulates over k from a +0.0 accumulator with fused
// multiply-adds.  The accumulator matters for signed zeros only: (-0.0)*r or 0.0*(-r) alone would be -0.0, dgemm gives
// +0.0 -- and the sign of a zero direction component decides which side of a box plane an in-plane ray is on.
LRC_DI double dgemm_row(double a, double b, double c, double r0, double r1, double r2) {
    return __builtin_fma(c, r2, __builtin_fma(b, r1, __builtin_fma(a, r0, 0.0)));
}

// Ray (pose, i) of a pose-batched scan: origin = float32(pose[:3,3]); direction = float32(dirs3[i] @ R^T).
// The reference forms the product with np.dot(directions, pose[:3,:3].T) in float64 (lidar/indoor_lidar.py:127-131),
// i.e. BLAS dgemm, whose kernels accumulate over k with fused multiply-adds: out_j = fma(c, R[j][2],
// fma(b, R[j][1], fma(a, R[j][0], +0))).  Reproduced here term for term (dgemm_row), so the float64 product -- and hence the float32
// direction -- is bit-identical for rotated poses too (checked against vectors captured from the reference at
// yaw 0.7, tests/golden/).  c = pose[:3,3] in float64.
LRC_DI void gen_ray(const double* poses16, const double* dirs3, uint64_t pose, uint64_t i, V3& o, V3& d,
                    double& cx, double& cy, double& cz) {
    const double* M = poses16 + pose * 16;
    const double* dv = dirs3 + i * 3;
    const double a = dv[0], b = dv[1], c = dv[2];
    d.x = (float)dgemm_row(a, b, c, M[0], M[1], M[2]);
    d.y = (float)dgemm_row(a, b, c, M[4], M[5], M[6]);
    d.z = (float)dgemm_row(a, b, c, M[8], M[9], M[10]);
    cx = M[3]; cy = M[7]; cz = M[11];
    o.x = (float)cx; o.y = (float)cy; o.z = (float)cz;
}

// Ray (pose, i) of the dual-axis sensor from its noisy scan angles (phi, theta), float64, drawn on the host from the
// seeded stream (lidar/indoor_lidar.py:270-272): d = (cos(theta)cos(phi), cos(theta)sin(phi), sin(theta)), rotated as
// the reference rotates it, ray by ray with numpy's un-fused (d0*R[:,0] + d1*R[:,1]) + d2*R[:,2] (:283-287), then
// narrowed to float32.  Opt-in path: the device's double sin/cos are not guaranteed to round like the host's libm.
LRC_DI void gen_ray_angles(const double* poses16, uint64_t pose, double phi, double theta, V3& o, V3& d,
                           double& cx, double& cy, double& cz) {
    const double* M = poses16 + pose * 16;
    const double ct = cos(theta);
    const double d0 = ct * cos(phi), d1 = ct * sin(phi), d2 = sin(theta);
    d.x = (float)((d0 * M[0] + d1 * M[1]) + d2 * M[2]);
    d.y = (float)((d0 * M[4] + d1 * M[5]) + d2 * M[6]);
    d.z = (float)((d0 * M[8] + d1 * M[9]) + d2 * M[10]);
    cx = M[3]; cy = M[7]; cz = M[11];
    o.x = (float)cx; o.y = (float)cy; o.z = (float)cz;
}

// p = o + (d/|d|)*t : numpy float32, one rounding per operation (reference: raycast_engine_cpu.py:57-62).
// h receives the normalised direction.
LRC_DI void hit_point(V3 o, V3 d, float t, V3& h, V3& pt) {
    const float nrm = __builtin_sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z);
    h.x = d.x / nrm; h.y = d.y / nrm; h.z = d.z / nrm;
    pt.x = o.x + h.x * t; pt.y = o.y + h.y * t; pt.z = o.z + h.z * t;
}

}  // namespace lrcdev

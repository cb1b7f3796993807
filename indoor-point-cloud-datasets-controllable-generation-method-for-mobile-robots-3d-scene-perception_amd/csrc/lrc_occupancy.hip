// lrc_occupancy.hip -- "does the robot's bounding cube contain a mesh vertex?" for many positions (gfx950).
//
// SURVEY.md section 8(f) row N2.  The reference's trajectory planner decides whether a grid point or a
// waypoint is free by testing EVERY mesh vertex against the robot's axis-aligned cube, one position at a time
// (trajectory/auto_trajectory_generator.py:219-238, called from :129-139 for the free-space grid and from
// :345-356 for every waypoint of every candidate): O(positions x vertices) numpy passes.  This kernel answers
// all positions at once.  Exact: float64, cube = [p - half, p + half] with inclusive comparisons, as the
// reference's `(v >= robot_min) & (v <= robot_max)`.
#include <hip/hip_runtime.h>

#include <new>
#include <string>

#include "../../include/lidarcast.h"

extern "C" int lrc_internal_fail(int code, const char* msg);
extern "C" int lrc_internal_ctx_device(const lrc_ctx* ctx);

namespace {

constexpr int kQ = 64;       // positions per block
constexpr int kVPT = 8;      // vertices per thread

#define O_HIP(call)                                                                             \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                            \
            return lrc_internal_fail(e__ == hipErrorOutOfMemory ? LRC_ERR_OOM : LRC_ERR_HIP,    \
                                     (std::string(#call) + ": " + hipGetErrorString(e__)).c_str()); \
        }                                                                                       \
    } while (0)

__global__ __launch_bounds__(256) void occupancy_kernel(const double* verts, uint64_t V, const double* pts,
                                                        uint64_t Q, double half, uint32_t* flags) {
    __shared__ double s_lo[kQ * 3], s_hi[kQ * 3];
    __shared__ uint32_t s_hit[kQ];
    const uint64_t q0 = (uint64_t)blockIdx.y * kQ;
    const uint32_t nq = (uint32_t)((Q - q0) < (uint64_t)kQ ? (Q - q0) : (uint64_t)kQ);
    if (threadIdx.x < kQ) s_hit[threadIdx.x] = 0;
    for (uint32_t k = threadIdx.x; k < nq * 3; k += 256) {
        const double c = pts[q0 * 3 + k];
        s_lo[k] = c - half;       // robot_min = point - robot_half_size
        s_hi[k] = c + half;       // robot_max = point + robot_half_size
    }
    __syncthreads();
    const uint64_t v0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * kVPT;
    for (int j = 0; j < kVPT; ++j) {
        const uint64_t v = v0 + j;
        if (v >= V) break;
        const double x = verts[3 * v], y = verts[3 * v + 1], z = verts[3 * v + 2];
        for (uint32_t k = 0; k < nq; ++k) {
            const bool in = (x >= s_lo[3 * k]) & (x <= s_hi[3 * k]) & (y >= s_lo[3 * k + 1]) & (y <= s_hi[3 * k + 1]) &
                            (z >= s_lo[3 * k + 2]) & (z <= s_hi[3 * k + 2]);
            if (in) s_hit[k] = 1u;      // benign race: every writer stores 1
        }
    }
    __syncthreads();
    if (threadIdx.x < nq && s_hit[threadIdx.x]) atomicOr(&flags[q0 + threadIdx.x], 1u);
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
};

}  // namespace

struct lrc_occ {
    int device = 0;
    double* d_verts = nullptr;
    uint64_t V = 0;
};

extern "C" {

int lrc_occ_destroy(lrc_occ* occ) {
    if (!occ) return LRC_OK;
    (void)hipSetDevice(occ->device);
    if (occ->d_verts) (void)hipFree(occ->d_verts);
    delete occ;
    return LRC_OK;
}

int lrc_occ_create(lrc_ctx* ctx, const double* verts3, uint64_t V, lrc_occ** out_occ) {
    if (!out_occ) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_occ_create: out_occ is NULL");
    *out_occ = nullptr;
    if (!ctx || (V && !verts3)) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_occ_create: NULL argument");
    lrc_occ* o = new (std::nothrow) lrc_occ();
    if (!o) return lrc_internal_fail(LRC_ERR_OOM, "lrc_occ_create: out of host memory");
    o->device = lrc_internal_ctx_device(ctx);
    o->V = V;
    if (hipSetDevice(o->device) != hipSuccess) { delete o; return lrc_internal_fail(LRC_ERR_HIP, "hipSetDevice failed"); }
    if (V) {
        if (hipMalloc((void**)&o->d_verts, V * 24) != hipSuccess) {
            delete o;
            return lrc_internal_fail(LRC_ERR_OOM, "lrc_occ_create: out of device memory");
        }
        if (hipMemcpy(o->d_verts, verts3, V * 24, hipMemcpyHostToDevice) != hipSuccess) {
            lrc_occ_destroy(o);
            return lrc_internal_fail(LRC_ERR_HIP, "lrc_occ_create: upload failed");
        }
    }
    *out_occ = o;
    return LRC_OK;
}

int lrc_occ_query(lrc_occ* occ, const double* points3, uint64_t Q, double half, uint8_t* out_flags) {
    if (!occ || (Q && (!points3 || !out_flags)))
        return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_occ_query: NULL argument");
    if (Q == 0) return LRC_OK;
    if (occ->V == 0) { for (uint64_t i = 0; i < Q; ++i) out_flags[i] = 0; return LRC_OK; }
    O_HIP(hipSetDevice(occ->device));
    Buf dp, df;
    O_HIP(hipMalloc(&dp.p, Q * 24));
    O_HIP(hipMalloc(&df.p, Q * 4));
    O_HIP(hipMemcpy(dp.p, points3, Q * 24, hipMemcpyHostToDevice));
    O_HIP(hipMemset(df.p, 0, Q * 4));
    const uint64_t gx = (occ->V + 256ull * kVPT - 1) / (256ull * kVPT), gy = (Q + kQ - 1) / kQ;
    if (gx > 0x7FFFFFFFull || gy > 65535ull) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_occ_query: too large");
    hipLaunchKernelGGL(occupancy_kernel, dim3((uint32_t)gx, (uint32_t)gy), dim3(256), 0, nullptr,
                       (const double*)occ->d_verts, occ->V, (const double*)dp.p, Q, half, (uint32_t*)df.p);
    O_HIP(hipGetLastError());
    O_HIP(hipDeviceSynchronize());
    uint32_t* tmp = new (std::nothrow) uint32_t[Q];
    if (!tmp) return lrc_internal_fail(LRC_ERR_OOM, "lrc_occ_query: out of host memory");
    hipError_t e = hipMemcpy(tmp, df.p, Q * 4, hipMemcpyDeviceToHost);
    for (uint64_t i = 0; i < Q; ++i) out_flags[i] = tmp[i] ? 1 : 0;
    delete[] tmp;
    if (e != hipSuccess) return lrc_internal_fail(LRC_ERR_HIP, "lrc_occ_query: download failed");
    return LRC_OK;
}

}  // extern "C"

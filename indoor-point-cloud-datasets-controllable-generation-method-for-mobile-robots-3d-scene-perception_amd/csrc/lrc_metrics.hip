// lrc_metrics.hip -- point-cloud similarity metrics of the validation stage (gfx950).
//
// SURVEY.md section 8(f) row N3.  The reference evaluates generated clouds with three sampled metrics
// (evaluate_single_scene.py:55-111): Chamfer = mean(min_j |x_i - y_j|) + mean(min_i |x_i - y_j|) on 5 000-point
// samples, Hausdorff = max of the two directed maxima on 3 000-point samples, and MMD with an RBF kernel
// exp(-gamma |x-y|^2) on up to 10 000-point samples -- all as dense numpy distance matrices.  The two kernels
// below are the O(n*m) parts:
//   min_dist_kernel : for every row of A the distance to the nearest row of B   (Chamfer, Hausdorff)
//   rbf_sum_kernel  : sum_ij exp(-gamma * max(|a_i|^2 + |b_j|^2 - 2 a_i.b_j, 0))  (one term of the MMD)
// B is streamed through LDS in tiles; float32 distances as numpy computes them for float32 clouds, float64
// accumulation of the kernel sum.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/lidarcast.h"

extern "C" int lrc_internal_fail(int code, const char* msg);
extern "C" int lrc_internal_ctx_device(const lrc_ctx* ctx);

namespace {

constexpr int kTile = 1024;

#define M_HIP(call)                                                                             \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                            \
            return lrc_internal_fail(e__ == hipErrorOutOfMemory ? LRC_ERR_OOM : LRC_ERR_HIP,    \
                                     (std::string(#call) + ": " + hipGetErrorString(e__)).c_str()); \
        }                                                                                       \
    } while (0)

__global__ __launch_bounds__(256) void min_dist_kernel(const float* A, uint64_t n, const float* B, uint64_t m,
                                                       float* out) {
    __shared__ float sb[kTile * 3];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (i < n) { ax = A[3 * i]; ay = A[3 * i + 1]; az = A[3 * i + 2]; }
    float best = __builtin_inff();
    for (uint64_t base = 0; base < m; base += kTile) {
        const uint32_t cnt = (uint32_t)((m - base) < (uint64_t)kTile ? (m - base) : (uint64_t)kTile);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cnt * 3; k += 256) sb[k] = B[3 * base + k];
        __syncthreads();
        for (uint32_t k = 0; k < cnt; ++k) {
            const float dx = ax - sb[3 * k], dy = ay - sb[3 * k + 1], dz = az - sb[3 * k + 2];
            const float d2 = (dx * dx + dy * dy) + dz * dz;     // numpy: sum of squares, then one sqrt
            best = d2 < best ? d2 : best;
        }
    }
    if (i < n) out[i] = __builtin_sqrtf(best);
}

__global__ __launch_bounds__(256) void rbf_sum_kernel(const float* A, uint64_t n, const float* B, uint64_t m,
                                                      double gamma, double* out_partial) {
    __shared__ float sb[kTile * 4];
    __shared__ double red[256];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    double ax = 0, ay = 0, az = 0, an = 0;
    if (i < n) {
        ax = A[3 * i]; ay = A[3 * i + 1]; az = A[3 * i + 2];
        an = (ax * ax + ay * ay) + az * az;
    }
    double acc = 0.0;
    for (uint64_t base = 0; base < m; base += kTile) {
        const uint32_t cnt = (uint32_t)((m - base) < (uint64_t)kTile ? (m - base) : (uint64_t)kTile);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cnt; k += 256) {
            const float x = B[3 * (base + k)], y = B[3 * (base + k) + 1], z = B[3 * (base + k) + 2];
            sb[4 * k] = x; sb[4 * k + 1] = y; sb[4 * k + 2] = z;
        }
        __syncthreads();
        if (i < n) {
            for (uint32_t k = 0; k < cnt; ++k) {
                const double bx = sb[4 * k], by = sb[4 * k + 1], bz = sb[4 * k + 2];
                const double bn = (bx * bx + by * by) + bz * bz;
                double d2 = an + bn - 2.0 * ((ax * bx + ay * by) + az * bz);
                d2 = d2 > 0.0 ? d2 : 0.0;
                acc += exp(-gamma * d2);
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_partial[blockIdx.x] = red[0];
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" {

int lrc_min_distances(lrc_ctx* ctx, const float* a3, uint64_t n, const float* b3, uint64_t m, float* out_min) {
    if (!ctx || (n && (!a3 || !out_min)) || (m && !b3))
        return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_min_distances: NULL argument");
    if (n == 0) return LRC_OK;
    if (m == 0) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_min_distances: the target cloud is empty");
    M_HIP(hipSetDevice(lrc_internal_ctx_device(ctx)));
    Buf da, db, dout;
    M_HIP(hipMalloc(&da.p, n * 12));
    M_HIP(hipMalloc(&db.p, m * 12));
    M_HIP(hipMalloc(&dout.p, n * 4));
    M_HIP(hipMemcpy(da.p, a3, n * 12, hipMemcpyHostToDevice));
    M_HIP(hipMemcpy(db.p, b3, m * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(min_dist_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, nullptr, (const float*)da.p, n,
                       (const float*)db.p, m, (float*)dout.p);
    M_HIP(hipGetLastError());
    M_HIP(hipDeviceSynchronize());
    M_HIP(hipMemcpy(out_min, dout.p, n * 4, hipMemcpyDeviceToHost));
    return LRC_OK;
}

int lrc_rbf_kernel_sum(lrc_ctx* ctx, const float* a3, uint64_t n, const float* b3, uint64_t m, double gamma,
                       double* out_sum) {
    if (!ctx || !out_sum || (n && !a3) || (m && !b3))
        return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_rbf_kernel_sum: NULL argument");
    *out_sum = 0.0;
    if (n == 0 || m == 0) return LRC_OK;
    M_HIP(hipSetDevice(lrc_internal_ctx_device(ctx)));
    const uint32_t nblk = (uint32_t)((n + 255) / 256);
    Buf da, db, dp;
    M_HIP(hipMalloc(&da.p, n * 12));
    M_HIP(hipMalloc(&db.p, m * 12));
    M_HIP(hipMalloc(&dp.p, (size_t)nblk * 8));
    M_HIP(hipMemcpy(da.p, a3, n * 12, hipMemcpyHostToDevice));
    M_HIP(hipMemcpy(db.p, b3, m * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rbf_sum_kernel, dim3(nblk), dim3(256), 0, nullptr, (const float*)da.p, n, (const float*)db.p, m,
                       gamma, (double*)dp.p);
    M_HIP(hipGetLastError());
    M_HIP(hipDeviceSynchronize());
    double* part = new (std::nothrow) double[nblk];
    if (!part) return lrc_internal_fail(LRC_ERR_OOM, "lrc_rbf_kernel_sum: out of host memory");
    hipError_t e = hipMemcpy(part, dp.p, (size_t)nblk * 8, hipMemcpyDeviceToHost);
    double s = 0.0;
    for (uint32_t k = 0; k < nblk; ++k) s += part[k];
    delete[] part;
    if (e != hipSuccess) return lrc_internal_fail(LRC_ERR_HIP, "lrc_rbf_kernel_sum: download failed");
    *out_sum = s;
    return LRC_OK;
}

}  // extern "C"

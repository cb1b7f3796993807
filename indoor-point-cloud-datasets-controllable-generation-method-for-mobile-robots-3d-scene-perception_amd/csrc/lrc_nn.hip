// lrc_nn.hip -- exact 1-nearest-neighbour lookup into an annotated point cloud (gfx950).
//
// SURVEY.md section 8(f) row N1.  The reference attaches colour / semantic / instance labels to the scan
// output at export time with  sklearn.neighbors.NearestNeighbors(n_neighbors=1, algorithm='ball_tree')
// fitted on the raw annotated S3DIS cloud and queried with every hit point
// (containers/s3dis_sim_scene.py:416-424).  This file provides the same query on the GPU:
//   * lrc_nn_create   : uniform grid over the annotated points (counting sort on the host, once per cloud)
//   * lrc_nn_query    : one lane per query point, expanding cube shells of cells until the best distance
//                       found cannot be beaten by any unsearched cell; float64 distances
//                       ((dx*dx + dy*dy) + dz*dz, the metric sklearn evaluates), ties -> smaller index.
// It is used two ways: at export time on the hit cloud (the reference's semantics, bit-identical indices),
// and once per mesh to bake per-triangle labels from triangle centroids, which the trace kernel then writes
// back per ray (lrc_hits.sem / .ins).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/lidarcast.h"

extern "C" int lrc_internal_fail(int code, const char* msg);     // lidarcast.hip: sets lrc_last_error()
extern "C" int lrc_internal_ctx_device(const lrc_ctx* ctx);

namespace {

#define NN_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                            \
            return lrc_internal_fail(e__ == hipErrorOutOfMemory ? LRC_ERR_OOM : LRC_ERR_HIP,    \
                                     (std::string(#call) + ": " + hipGetErrorString(e__)).c_str()); \
        }                                                                                       \
    } while (0)

struct GridDev {
    const double* pts;        // (M,3) points sorted by cell
    const uint32_t* orig;     // (M) original row of each sorted point
    const uint32_t* start;    // (ncells+1) first sorted point of each cell
    double lo[3];
    double h, inv_h;
    int n[3];
    uint32_t M;
};

__device__ __forceinline__ int cell_of(double x, double lo, double inv_h, int n) {
    const double f = floor((x - lo) * inv_h);
    int c = f < 0.0 ? 0 : (f >= (double)n ? n - 1 : (int)f);
    return c;
}

__global__ __launch_bounds__(256) void nn_query_kernel(const GridDev g, const float* q3, uint64_t K,
                                                       uint32_t* out_idx, double* out_dist) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    const double qx = (double)q3[3 * i], qy = (double)q3[3 * i + 1], qz = (double)q3[3 * i + 2];
    const int cx = cell_of(qx, g.lo[0], g.inv_h, g.n[0]);
    const int cy = cell_of(qy, g.lo[1], g.inv_h, g.n[1]);
    const int cz = cell_of(qz, g.lo[2], g.inv_h, g.n[2]);
    double best = INFINITY;       // squared distance
    uint32_t best_i = 0xFFFFFFFFu;
    const int rmax = max(max(max(cx, g.n[0] - 1 - cx), max(cy, g.n[1] - 1 - cy)), max(cz, g.n[2] - 1 - cz));
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.n[2] - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, g.n[1] - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.n[0] - 1);
        for (int z = z0; z <= z1; ++z) {
            const bool zface = (z == cz - r) | (z == cz + r);
            for (int y = y0; y <= y1; ++y) {
                const bool yface = (y == cy - r) | (y == cy + r);
                // on a z- or y-face of the shell the whole x-run belongs to it; otherwise only the two ends
                const int step = (zface | yface) ? 1 : max(2 * r, 1);
                for (int x = cx - r; x <= cx + r; x += step) {
                    if (x < x0 || x > x1) continue;
                    const size_t c = ((size_t)z * g.n[1] + y) * g.n[0] + x;
                    const uint32_t a = g.start[c], b = g.start[c + 1];
                    for (uint32_t k = a; k < b; ++k) {
                        const double dx = qx - g.pts[3 * (size_t)k], dy = qy - g.pts[3 * (size_t)k + 1],
                                     dz = qz - g.pts[3 * (size_t)k + 2];
                        const double d2 = (dx * dx + dy * dy) + dz * dz;
                        const uint32_t oi = g.orig[k];
                        if (d2 < best || (d2 == best && oi < best_i)) { best = d2; best_i = oi; }
                    }
                }
            }
        }
        if (best_i != 0xFFFFFFFFu) {
            // everything inside the cube of cells [c-r, c+r]^3 has been searched: a closer point would have
            // to lie outside it, i.e. farther than the distance from q to the nearest face of that cube
            const double mx = fmin(qx - (g.lo[0] + (double)(cx - r) * g.h), (g.lo[0] + (double)(cx + r + 1) * g.h) - qx);
            const double my = fmin(qy - (g.lo[1] + (double)(cy - r) * g.h), (g.lo[1] + (double)(cy + r + 1) * g.h) - qy);
            const double mz = fmin(qz - (g.lo[2] + (double)(cz - r) * g.h), (g.lo[2] + (double)(cz + r + 1) * g.h) - qz);
            // (shrunk by 1e-6 cell: a point binned by floor((p-lo)/h) may sit one rounding below its cell's edge)
            const double margin = fmin(mx, fmin(my, mz)) - 1.0e-6 * g.h;
            if (margin > 0.0 && best <= margin * margin) break;
        }
    }
    out_idx[i] = best_i;
    if (out_dist) out_dist[i] = sqrt(best);
}

}  // namespace

struct lrc_nn {
    lrc_ctx* ctx = nullptr;
    int device = 0;
    double* d_pts = nullptr;
    uint32_t* d_orig = nullptr;
    uint32_t* d_start = nullptr;
    GridDev g{};
    uint64_t M = 0, ncells = 0;
};

extern "C" {

int lrc_nn_destroy(lrc_nn* nn) {
    if (!nn) return LRC_OK;
    (void)hipSetDevice(nn->device);
    if (nn->d_pts) (void)hipFree(nn->d_pts);
    if (nn->d_orig) (void)hipFree(nn->d_orig);
    if (nn->d_start) (void)hipFree(nn->d_start);
    delete nn;
    return LRC_OK;
}

int lrc_nn_create(lrc_ctx* ctx, const double* points3, uint64_t M, double cell_size, lrc_nn** out_nn) {
    if (!out_nn) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_nn_create: out_nn is NULL");
    *out_nn = nullptr;
    if (!ctx || !points3) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_nn_create: NULL context or points");
    if (M == 0 || M >= 0xFFFFFFFFull)
        return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_nn_create: need 1 <= M < 2^32 - 1 points");
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint64_t i = 0; i < M; ++i)
        for (int k = 0; k < 3; ++k) {
            const double v = points3[3 * i + k];
            if (!std::isfinite(v)) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_nn_create: non-finite point");
            lo[k] = std::min(lo[k], v);
            hi[k] = std::max(hi[k], v);
        }
    double ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    double h = cell_size;
    if (!(h > 0.0)) {
        // surfaces, not volumes: aim at a few points per occupied cell of a 2-D sheet
        const double area = std::max({ext[0] * ext[1], ext[1] * ext[2], ext[0] * ext[2], 1e-12});
        h = std::sqrt(area / (double)M) * 2.0;
    }
    const double emax = std::max({ext[0], ext[1], ext[2], 1e-9});
    h = std::max(h, emax / 1024.0);                       // at most 1024 cells per axis
    int n[3];
    for (;;) {
        double cells = 1.0;
        for (int k = 0; k < 3; ++k) { n[k] = std::max(1, (int)std::floor(ext[k] / h) + 1); cells *= n[k]; }
        if (cells <= 64.0e6) break;
        h *= 1.26;
    }
    const uint64_t ncells = (uint64_t)n[0] * n[1] * n[2];
    const double inv_h = 1.0 / h;

    std::vector<uint32_t> cell(M), start(ncells + 1, 0), orig(M);
    std::vector<double> sorted(3 * M);
    try {
        for (uint64_t i = 0; i < M; ++i) {
            uint64_t c[3];
            for (int k = 0; k < 3; ++k) {
                double f = std::floor((points3[3 * i + k] - lo[k]) * inv_h);
                c[k] = f < 0 ? 0 : (f >= n[k] ? (uint64_t)n[k] - 1 : (uint64_t)f);
            }
            cell[i] = (uint32_t)((c[2] * n[1] + c[1]) * n[0] + c[0]);
            ++start[cell[i] + 1];
        }
        for (uint64_t c = 0; c < ncells; ++c) start[c + 1] += start[c];
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        for (uint64_t i = 0; i < M; ++i) {          // stable: equal cells keep ascending original order
            const uint32_t d = fill[cell[i]]++;
            orig[d] = (uint32_t)i;
            std::memcpy(&sorted[3 * (size_t)d], &points3[3 * i], 24);
        }
    } catch (const std::bad_alloc&) {
        return lrc_internal_fail(LRC_ERR_OOM, "lrc_nn_create: out of host memory");
    }

    lrc_nn* nn = new (std::nothrow) lrc_nn();
    if (!nn) return lrc_internal_fail(LRC_ERR_OOM, "lrc_nn_create: out of host memory");
    nn->ctx = ctx;
    nn->device = lrc_internal_ctx_device(ctx);
    nn->M = M;
    nn->ncells = ncells;
    auto bail = [&](int rc) { lrc_nn_destroy(nn); return rc; };
    if (hipSetDevice(nn->device) != hipSuccess) return bail(lrc_internal_fail(LRC_ERR_HIP, "hipSetDevice failed"));
    if (hipMalloc((void**)&nn->d_pts, M * 24) != hipSuccess || hipMalloc((void**)&nn->d_orig, M * 4) != hipSuccess ||
        hipMalloc((void**)&nn->d_start, (ncells + 1) * 4) != hipSuccess)
        return bail(lrc_internal_fail(LRC_ERR_OOM, "lrc_nn_create: out of device memory"));
    if (hipMemcpy(nn->d_pts, sorted.data(), M * 24, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(nn->d_orig, orig.data(), M * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(nn->d_start, start.data(), (ncells + 1) * 4, hipMemcpyHostToDevice) != hipSuccess)
        return bail(lrc_internal_fail(LRC_ERR_HIP, "lrc_nn_create: upload failed"));
    GridDev& g = nn->g;
    g.pts = nn->d_pts; g.orig = nn->d_orig; g.start = nn->d_start;
    for (int k = 0; k < 3; ++k) { g.lo[k] = lo[k]; g.n[k] = n[k]; }
    g.h = h; g.inv_h = inv_h; g.M = (uint32_t)M;
    *out_nn = nn;
    return LRC_OK;
}

int lrc_nn_query_dev(lrc_nn* nn, const float* d_query3, uint64_t K, uint32_t* d_out_index,
                     double* d_out_dist, void* stream) {
    if (!nn || (K && (!d_query3 || !d_out_index)))
        return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_nn_query_dev: NULL argument");
    if (K == 0) return LRC_OK;
    NN_HIP(hipSetDevice(nn->device));
    const uint64_t nblk = (K + 255) / 256;
    if (nblk > 0x7FFFFFFFull) return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_nn_query_dev: too many queries");
    hipLaunchKernelGGL(nn_query_kernel, dim3((uint32_t)nblk), dim3(256), 0, (hipStream_t)stream, nn->g, d_query3, K,
                       d_out_index, d_out_dist);
    NN_HIP(hipGetLastError());
    return LRC_OK;
}

int lrc_nn_query(lrc_nn* nn, const float* query3, uint64_t K, uint32_t* out_index, double* out_dist) {
    if (!nn || (K && (!query3 || !out_index)))
        return lrc_internal_fail(LRC_ERR_INVALID_ARG, "lrc_nn_query: NULL argument");
    if (K == 0) return LRC_OK;
    NN_HIP(hipSetDevice(nn->device));
    float* dq = nullptr; uint32_t* di = nullptr; double* dd = nullptr;
    int rc = LRC_OK;
    if (hipMalloc((void**)&dq, K * 12) != hipSuccess || hipMalloc((void**)&di, K * 4) != hipSuccess ||
        (out_dist && hipMalloc((void**)&dd, K * 8) != hipSuccess)) {
        rc = lrc_internal_fail(LRC_ERR_OOM, "lrc_nn_query: out of device memory");
    } else if (hipMemcpy(dq, query3, K * 12, hipMemcpyHostToDevice) != hipSuccess) {
        rc = lrc_internal_fail(LRC_ERR_HIP, "lrc_nn_query: upload failed");
    } else if ((rc = lrc_nn_query_dev(nn, dq, K, di, dd, nullptr)) == LRC_OK) {
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out_index, di, K * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            (out_dist && hipMemcpy(out_dist, dd, K * 8, hipMemcpyDeviceToHost) != hipSuccess))
            rc = lrc_internal_fail(LRC_ERR_HIP, "lrc_nn_query: kernel or download failed");
    }
    if (dq) (void)hipFree(dq);
    if (di) (void)hipFree(di);
    if (dd) (void)hipFree(dd);
    return rc;
}

}  // extern "C"

// lidarcast.hip -- HIP kernels and C ABI of liblidarcast (gfx950 / MI355X only).
//
// Implements include/lidarcast.h.  Replaces, behind the reference's engine boundary
// (raycast_engine/raycast_engine.py:16-61), the Open3D/Embree work of
// raycast_engine/raycast_engine_cpu.py:46-73 and the numpy post-processing of :95-107,
// and, for pose batches, the per-waypoint loop body of s3dis_simulator.py:254-264.
//
// Kernels
//   trace_kernel<GEN,..> one lane per ray, one wave per workgroup; while-while BVH2 traversal with the per-lane
//                        stack in LDS ([depth][lane], conflict free); wave-uniform nodes fetched through the
//                        scalar cache (s_load); fused hit write-back (t, prim, normal, point, labels, range
//                        filter, incident angle, packed (t,label) pair, per-wave keep count).
//                        GEN = rays generated from (pose, direction table) inside the kernel.
//                        QN = per-lane node fetches from the 32-byte quantised node images (15-bit grid per axis,
//                        one v_perm_b32 per plane); float32 world-space nodes for the rays the grid is not proven for.
//   compact_*            stable stream compaction of the fixed-stride records into frame order.
//   cloud_*              the same compaction with the hit point rebuilt from (t, label) pairs (multi-GPU assembly).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/lidarcast.h"
#include "lrc_bvh.h"
#include "lrc_bvh_device.h"
#include "lrc_qnodes.h"
#include "lrc_device.h"

using namespace lrcdev;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define LRC_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            (void)hipGetLastError();                                                       \
            return fail(e__ == hipErrorOutOfMemory ? LRC_ERR_OOM : LRC_ERR_HIP,            \
                        std::string(#call) + ": " + hipGetErrorString(e__));               \
        }                                                                                  \
    } while (0)

constexpr int kBlock = 256;          // compaction kernels
#ifndef LRC_TRACE_BLOCK
#define LRC_TRACE_BLOCK 64
#endif
constexpr int kTBlock = LRC_TRACE_BLOCK;   // trace kernel workgroup (rays per tile)

}  // namespace

enum PoolSlot { kPoolRays = 0, kPoolT, kPoolPrim, kPoolNormal, kPoolPoint, kPoolSem, kPoolIns, kPoolInc, kPoolInten,
                kPoolPoses, kPoolDirs, kPoolOffs, kPoolCen, kPoolNoise,
                // *_compact entry points: scan angles in, per-wave keep counts, compacted frame arrays out
                kPoolAngles, kPoolKeep, kPoolTile, kPoolCounts, kPoolOutPoint, kPoolOutSem, kPoolOutIns, kPoolOutInc,
                kPoolOutIdx, kPoolOutXyzl, kPoolOutRange, kPoolStats, kPoolFrameStats, kPoolSlots };

// triangle records fetched per leaf round trip by the product kernels.  With edge records the pair fits 64 VGPRs, i.e. 8
// waves per SIMD: -2...-4 % trace time on all benchmark scenes against one record per round trip.
constexpr int kLeafW = 2;
#ifndef LRC_REBUILD_R
#define LRC_REBUILD_R 2      // tiles (of 64 entries) one wave of the cloud rebuild handles (1, 2, 4, 8 measured equal)
#endif

struct lrc_ctx {
    int device = 0;
    // staging buffers of the host-pointer entry points, grown on demand and reused (a per-waypoint caller
    // such as the reference loop, s3dis_simulator.py:254-264, would otherwise pay hipMalloc/hipFree per pose)
    void* pool[kPoolSlots] = {};
    size_t pool_cap[kPoolSlots] = {};
    // compaction scratch (grown on demand, reused): per-tile counts and exclusive offsets.  Two sets: the
    // cloud rebuild of the multi-GPU path runs on its own stream next to the compaction of the local scan.
    struct TileScratch {
        uint32_t* d_tile_off = nullptr;     // offset of a tile inside its super tile (1024 tiles)
        uint32_t* d_tile_cnt = nullptr;
        uint32_t* d_super_total = nullptr;  // kept entries per super tile
        uint64_t* d_super_base = nullptr;   // exclusive prefix of d_super_total (+ grand total)
        uint64_t tile_cap = 0;
        double* d_dirs_soa = nullptr;       // direction table transposed to x[N] y[N] z[N] (cloud rebuild only)
        uint64_t dirs_cap = 0;
    };
    TileScratch cloud_scratch;
    // lrc_compact_dev: one scratch set per caller stream (a caller that keeps two scans in flight on two streams compacts on
    // both); a set handed on to another stream is first ordered behind its last use (scratch_for)
    static constexpr int kCompactSets = 4;
    TileScratch compact_scratch[kCompactSets];
    hipStream_t compact_stream[kCompactSets] = {};
    hipEvent_t compact_done[kCompactSets] = {};
    bool compact_used[kCompactSets] = {};
    int compact_next = 0;
    // Dispatch chaining of trace launches ("the launch tail"): the LAST workgroup of every trace launch writes the
    // launch's sequence number to this signal word at its first instruction; a trace launch on ANOTHER stream than the
    // previous one is held behind hipStreamWaitValue64(word >= previous sequence number), i.e. it starts the moment the
    // previous launch has handed out its last workgroup -- its waves fill the slots the previous launch's tail leaves empty,
    // and launches that a caller keeps in flight on two streams run staggered instead of falling into phase.
    uint64_t* chain_word = nullptr;     // hipMallocSignalMemory; NULL: not supported here, launches are never chained
    uint64_t chain_seq = 0;             // sequence number of the last chained trace launch
    hipStream_t chain_stream = nullptr; // ... and the stream it went to
    bool chain_enabled = false;         // lrc_ctx_set_launch_chaining (opt-in: measured equal to what the dispatcher does itself)
    // *_compact entry points: kernels on one stream, the transfers of finished pose chunks on another
    hipStream_t s_compute = nullptr, s_copy = nullptr, s_stats = nullptr;
    hipEvent_t ev_chunk[8] = {}, ev_compact[8] = {};
    uint64_t* h_counts = nullptr;       // page-locked landing area of the per-pose counts and statistics (async copies
    uint64_t h_counts_cap = 0;          // need one): counts (P u64) | 4 x P doubles of per-pose statistics
    lrc::DeviceArena build_arena;       // scratch of the device scene build, reused from scene to scene
    float* stat_scratch = nullptr;      // chunk sums of lrc_cloud_range_stats_dev (calls of one context must not overlap
    uint64_t stat_scratch_cap = 0;      // on different streams: handles are not thread-safe)
};

struct lrc_table {            // a sensor's direction table resident in HBM (lrc_table_create)
    lrc_ctx* ctx = nullptr;
    double* d_dirs3 = nullptr;
    uint64_t n = 0;
};

struct lrc_scene {
    lrc_ctx* ctx = nullptr;
    void* slab = nullptr;             // device-built scenes: ONE allocation holds every array below except d_slot_sphere / *4
    float4* d_nodes = nullptr;
    float4* d_tris = nullptr;
    uint32_t* d_slot_prim = nullptr;
    uint32_t* d_slot_label = nullptr;
    float* d_slot_box = nullptr;      // per leaf slot the triangle's exact vertex box (lo xyz, hi xyz)
    float4* d_prim_plane = nullptr;   // per caller's triangle row: (v0, label bits), (Ng, 0): lrc_cloud_from_prims_dev
    std::mutex plane_mutex;           // ... built on first use, published complete (ensure_prim_plane)
    float4* d_slot_sphere = nullptr;  // per leaf slot: centre of the triangle's box + bounding radius (sector_kernel)
    // quantised node images of the SAME tree ("32-byte nodes"): child boxes on a 15-bit grid
    // per axis, rounded outward (margin 1/16 cell).  d_nodes_q: 32 B per node for the per-lane fetches; d_nodes_n: the same
    // boxes as normalised float32 (64 B per node) for the scalar fetches.  NULL when the grid does not fit the scene.
    uint4* d_nodes_q = nullptr;
    float4* d_nodes_n = nullptr;
    // the same tree collapsed to four children per node (every second level removed), on the same grid:
    // d_nodes_q4 64 B per node (per child lo|hi<<16 x, y, z + reference), d_nodes_n4 128 B (per child lo, hi, ref, pad)
    uint4* d_nodes_q4 = nullptr;
    float4* d_nodes_n4 = nullptr;
    uint64_t num_nodes4 = 0;
    float qbase[3] = {0, 0, 0}, qW[3] = {1, 1, 1}, qinvW[3] = {1, 1, 1};
    const lrc_grid* cur_grid = nullptr;   // set around a grid scan (launch_trace gen == 3)
    lrc_scene_info info{};
    lrc_scan_options opts{};          // sticky opt-in options (lrc_scene_set_options)
    uint64_t launches = 0, rays = 0;
};

// ------------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------------
namespace {

// A cloud rebuild from gathered triangle ids (lrc_cloud_from_prims_dev), as a kernel argument block.
struct RebuildParams {
    const double* __restrict__ poses16;
    const double* __restrict__ dirs_soa;     // direction table transposed to x[N] y[N] z[N]
    const uint32_t* __restrict__ prims;      // entry (pose p, ray i): slab p / pps, word (p % pps) * seg_len + i
    uint32_t pps;
    uint64_t stride;                         // words between slabs
    uint32_t seg_len, tps, ntiles, nseg;
    const float4* __restrict__ plane;
    uint32_t num_prims;
    const uint32_t* __restrict__ tile_off;
    const uint64_t* __restrict__ super_base;
    float4* __restrict__ out_xyzl;
    uint64_t* __restrict__ counts;
    uint32_t skip_slab;                      // tiles of this slab are not rebuilt (the caller scatters its own records); ~0u: none
};

struct TraceParams {
    const float4* nodes;
    const float4* tris;
    const uint32_t* slot_prim;
    const uint32_t* slot_label;
    const float* slot_box;     // per slot lo xyz, hi xyz of the triangle's vertices
    const float4* prim_plane;  // per caller's triangle row: (v0, label bits), (Ng, 0)
    uint32_t num_nodes;
    const uint4* nodes_q;      // QN kernels: 32-byte quantised nodes (per-lane fetches) ...
    const float4* nodes_n;     // ... and the same boxes as normalised float32 (scalar fetches)
    const uint4* nodes_q4;     // QN = 2: the tree collapsed to four children per node, 64 B per node ...
    const float4* nodes_n4;    // ... and 128 B per node as normalised float32
    uint32_t stack_cap;        // entries of the per-lane LDS stack
    uint32_t tile_chunk_log2;  // 0: eight contiguous tile ranges, one per XCD; k + 1: chunks of 2^k tiles round robin
    uint32_t force_redo;       // test hook (LRC_DEBUG_FORCE_REDO=m): rays with gid % m == 0 take the redo path as well
    uint64_t* chain_word;      // launch chain (lrc_ctx::chain_word): the last workgroup stores chain_seq here when it starts
    uint64_t chain_seq;
    // scan pipeline (lrc_pipe): the first pre.blocks workgroups of the launch do not trace -- each compacts one 64-entry
    // tile of an EARLIER scan's records (whose offsets a scan pass has tabulated).  They are handed out first, i.e. into the
    // wave slots the previous launch's long last waves leave empty, and are gone in microseconds; the trace tiles follow in
    // the same grid with no barrier in between.
    struct Pre {
        uint32_t blocks;           // leading workgroups in all; each scatter workgroup takes kPreTiles tiles
        uint32_t rows_only;        // only the packed (x, y, z, label) rows (+ counts) are asked for: the batched form
        // sharded form (lrc_pipe_submit_sharded): the assembly of an EARLIER scan of all ranks rides here -- the first
        // own_blocks workgroups scatter this rank's own rows from its records at the offsets of the assembled cloud
        // (tile_base, super_base: the scan over ALL ranks' keep counts), the others rebuild the other ranks' rows from their
        // gathered triangle ids (rebuild_tiles, LRC_REBUILD_R tiles each; tiles of the own slab are skipped)
        uint32_t sharded, own_blocks;
        uint64_t tile_base;
        const uint64_t* super_base;
        RebuildParams rq;
        uint64_t seg_len, tps, ntiles, nseg;
        const uint32_t* tile_off;
        const uint32_t* super_total;
        lrc_compact_io io;
    } pre;
    float qbase[3], qW[3], qinvW[3];   // normalised coordinate n = (x - qbase) * qinvW in [2, 4); qW = 1 / qinvW = 2^k
    // inputs
    const float* rays6;        // explicit rays (GEN = false)
    const uint64_t* seg_offsets;   // explicit rays in S segments (poses): (S+1) ray offsets, or NULL
    const double* seg_centers3;    // (S,3) range-filter centres of the segments
    uint32_t num_segments;
    const double* poses16;     // GEN = 1, 2
    const double* dirs3;       // GEN = 1: (N,3) sensor-frame direction table
    const double* angles2;     // GEN = 2: (P*N,2) noisy (phi, theta) of the dual-axis sensor
    const uint8_t* keep_mask;  // GEN = 2, nullable: 0 = ray dropped by the sensor (never cast, reported as a miss)
    uint32_t* stats;           // STATS build only: kStatsWords counters per ray
    uint64_t rays_per_pose;
    uint64_t total;
    int has_center;
    double cx, cy, cz;
    double max_range;
    // opt-in sensor-realism options (lrc_scan_options; all off by default = the reference's behaviour)
    double min_range;
    const float* range_noise;
    int incident_mode;
    lrc_hits out;
};

// workgroup -> tile remap: consecutive tiles land on the same XCD (blocks b, b+8, ... share an L2),
// so each XCD's 4 MiB L2 keeps the part of the scene its run of poses/scanlines looks at.
// klog > 0: the tiles are dealt to the XCDs in chunks of 2^(klog-1) consecutive tiles, round robin (chunk c -> XCD c % 8),
// instead of eight contiguous ranges.  Tiles beyond the last full round of eight chunks keep their own number.  Shifts
// and masks of wave-uniform values only: the scalar unit does it.
__device__ __forceinline__ uint32_t xcd_tile_chunked(uint32_t b, uint32_t nwg, uint32_t klog) {
    const uint32_t k = klog - 1u;
    const uint32_t full = (nwg >> (k + 3u)) << (k + 3u);
    if (b >= full) return b;
    const uint32_t x = b & 7u, j = b >> 3;
    return ((((j >> k) << 3) + x) << k) | (j & ((1u << k) - 1u));
}

__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t nwg) {
#ifdef LRC_NO_XCD_REMAP
    return b;      // A/B build only
#endif
    uint32_t q = nwg >> 3, r = nwg & 7u, x = b & 7u;
    uint32_t base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (b >> 3);
}

struct alignas(16) F4 { float x, y, z, w; };                       // plain 16-byte record (loads as dwordx4)
typedef __attribute__((address_space(4))) const float cfloat;      // constant address space: uniform loads become s_load
__device__ __forceinline__ F4 ld_uniform(const float4* gp) {
    cfloat* c = (cfloat*)gp;
    return F4{c[0], c[1], c[2], c[3]};
}

// ---- cloud rebuild from triangle ids: the device part ----
// R independent entries per lane: all ids are fetched first, then all plane records, then the arithmetic -- a chain
// of dependent gathers wants loads in flight per wave.  plane: 2 x float4 per triangle row: (v0, label bits), (Ng, 0)
// -- all a known hit needs to give t again.  Tiles [tile0, tile0 + R) below `limit` are processed.
template <int R>
__device__ __forceinline__ void rebuild_tiles(const RebuildParams& q, uint32_t tile0, uint32_t limit, uint32_t lane) {
    uint32_t seg[R], idx[R], prim[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint32_t tile = tile0 + k;
        prim[k] = LRC_INVALID_PRIM;
        seg[k] = 0; idx[k] = 0;
        if (tile < limit) {
            seg[k] = tile / q.tps;
            const uint32_t chunk = tile - seg[k] * q.tps, sb = seg[k] / q.pps;
            idx[k] = chunk * 64u + lane;
            if (idx[k] < q.seg_len && sb != q.skip_slab)     // read once: streaming load
                prim[k] = __builtin_nontemporal_load(q.prims + (uint64_t)sb * q.stride +
                                                     (uint64_t)(seg[k] - sb * q.pps) * q.seg_len + idx[k]);
        }
    }
    float4 pa[R], pb[R];
    double da[R], db[R], dc[R];
    uint64_t dst[R];
    bool keep[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        keep[k] = prim[k] != LRC_INVALID_PRIM;
        const unsigned long long m = __ballot(keep[k]);
        pa[k] = pb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        da[k] = db[k] = dc[k] = 0.0;
        dst[k] = 0;
        if (keep[k]) {
            const uint32_t tile = tile0 + k;
            // the tile's offset is wave-uniform too (written by the scan kernel before this launch): scalar loads
            typedef __attribute__((address_space(4))) const uint64_t cu64;
            typedef __attribute__((address_space(4))) const uint32_t cu32;
            dst[k] = ((cu64*)q.super_base)[tile >> 10] + ((cu32*)q.tile_off)[tile] + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
            da[k] = q.dirs_soa[idx[k]];
            db[k] = q.dirs_soa[(size_t)q.seg_len + idx[k]];
            dc[k] = q.dirs_soa[2 * (size_t)q.seg_len + idx[k]];
        }
        if (keep[k] && prim[k] < q.num_prims) {   // an id that is not a triangle of this scene: zero plane, never a wild read
            pa[k] = q.plane[(size_t)prim[k] * 2];
            pb[k] = q.plane[(size_t)prim[k] * 2 + 1];
        }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (!keep[k]) continue;
        // the tile's pose is wave-uniform: its matrix comes through the scalar cache (constant address space), not as twelve
        // 64-lane loads of one address through the vector memory path, which is what bounds this kernel
        typedef __attribute__((address_space(4))) const double cdouble;
        cdouble* M = (cdouble*)(q.poses16 + (size_t)seg[k] * 16);
        // gen_ray, with the direction already in registers (same FMA chain, lrc_device.h)
        V3 o, d, h, pt;
        d.x = (float)dgemm_row(da[k], db[k], dc[k], M[0], M[1], M[2]);
        d.y = (float)dgemm_row(da[k], db[k], dc[k], M[4], M[5], M[6]);
        d.z = (float)dgemm_row(da[k], db[k], dc[k], M[8], M[9], M[10]);
        o.x = (float)M[3]; o.y = (float)M[7]; o.z = (float)M[11];
        // the sender's scan established that this ray hits this triangle; t is the expression tri_hit evaluates
        // (T/|den| with T = Ng.(v0-O), den = Ng.D, sign-corrected), so v0 and Ng are all that is needed
        const V3 v0{pa[k].x, pa[k].y, pa[k].z}, ng{pb[k].x, pb[k].y, pb[k].z};
        const float den = dot3(ng, d);
        const uint32_t sgn = __float_as_uint(den) & 0x80000000u;
        const float t = xorsign(dot3(ng, sub3(v0, o)), sgn) / __builtin_fabsf(den);
        hit_point(o, d, t, h, pt);
        // streaming store: the rows are never read back here, and must not push the plane table out of the L2
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f row = {pt.x, pt.y, pt.z, pa[k].w};
        __builtin_nontemporal_store(row, (v4f*)(q.out_xyzl + dst[k]));
    }
}

// per-pose (segment) row counts from the scanned tile offsets: thread sg writes counts[sg]
__device__ __forceinline__ void rebuild_counts(const RebuildParams& q, uint32_t first, uint32_t step) {
    if (!q.counts) return;
    const uint32_t nsuper = (q.ntiles + 1023u) / 1024u;
    for (uint32_t sg = first; sg < q.nseg; sg += step) {
        const uint64_t g0 = (uint64_t)sg * q.tps, g1 = g0 + q.tps;
        const uint64_t o0 = g0 >= q.ntiles ? q.super_base[nsuper] : q.super_base[g0 >> 10] + q.tile_off[g0];
        const uint64_t o1 = g1 >= q.ntiles ? q.super_base[nsuper] : q.super_base[g1 >> 10] + q.tile_off[g1];
        q.counts[sg] = o1 - o0;
    }
}

// ---- fused write-back: everything after the closest hit is known (shared by the trace kernels) ------------------
// best_slot = 0xFFFFFFFF: no hit.  FILTER: apply the max_range filter (scans and casts with a centre).
// BY_PRIM: `best_slot` is the caller's triangle ROW (sector_kernel keys rays by (t, row)); labels and the normal then
// come from the per-row plane table instead of the per-slot arrays -- same values.
template <bool FILTER_ALWAYS, bool BY_PRIM = false>
__device__ __forceinline__ void write_back(const TraceParams& p, uint64_t gid, uint32_t tid, V3 o, V3 d, double cx,
                                           double cy, double cz, float tbest, uint32_t best_slot) {
    constexpr int GEN = FILTER_ALWAYS ? 1 : 0;
    bool keep = best_slot != 0xFFFFFFFFu;
    float t_out = __builtin_inff();
    uint32_t prim = LRC_INVALID_PRIM;
    float nx = 0.f, ny = 0.f, nz = 0.f, px = 0.f, py = 0.f, pz = 0.f;
    uint32_t label = 0;
    double inc = 0.0;
    float inten = 0.f;
    if (keep && p.range_noise) {
        // opt-in range noise (the reference declares range_noise_std but never applies it, lidar_intrinsics.py:
        // 364-389 has no caller): host-drawn additive noise on the range; a non-positive range drops the return
        tbest = tbest + p.range_noise[gid];
        keep = tbest > 0.0f;
    }
    if (keep) {
        // p = o + (d/|d|)*t : numpy float32, one rounding per operation (raycast_engine_cpu.py:57-62)
        V3 hh, pp;
        hit_point(o, d, tbest, hh, pp);
        const float hx = hh.x, hy = hh.y, hz = hh.z;
        px = pp.x; py = pp.y; pz = pp.z;
        // range filter + incident angle in float64 (raycast_engine_cpu.py:95-107)
        const double ex = (double)px - cx, ey = (double)py - cy, ez = (double)pz - cz;
        const double dist = __builtin_sqrt((ex * ex + ey * ey) + ez * ez);
        if (p.has_center || GEN != 0) keep = dist < p.max_range;
        if (p.min_range > 0.0) keep = keep & (dist >= p.min_range);     // opt-in; the reference never applies it
        if (keep) {
            t_out = tbest;
            if (BY_PRIM) {
                prim = best_slot;
                label = __float_as_uint(p.prim_plane[(size_t)best_slot * 2].w);
            } else {
                prim = p.slot_prim[best_slot];
                label = p.slot_label[best_slot];
            }
            if (p.out.normal3 || p.out.intensity || (p.out.incident_deg && p.incident_mode == 1)) {
                float4 c;
                if (BY_PRIM) { const float4 g = p.prim_plane[(size_t)best_slot * 2 + 1]; c = make_float4(0.f, g.x, g.y, g.z); }
                else c = p.tris[(size_t)best_slot * 3 + 2];
                const float len = __builtin_sqrtf(fma_(c.w, c.w, fma_(c.z, c.z, c.y * c.y)));
                nx = c.y / len; ny = c.z / len; nz = c.w / len;
            }
            if (p.out.incident_deg) {
                if (p.incident_mode == 1) {   // opt-in: angle between the ray and the surface normal
                    const double cs = __builtin_fabs(((double)hx * (double)nx + (double)hy * (double)ny) +
                                                     (double)hz * (double)nz);
                    inc = acos(cs < 1.0 ? cs : 1.0) * kRadToDeg;
                } else {
                    inc = acos(__builtin_fabs(ez / dist)) * kRadToDeg;
                }
            }
            if (p.out.intensity)      // opt-in: Lambertian return |h.n|, float32 (same FMA order as every dot product here)
                inten = __builtin_fabsf(fma_(hz, nz, fma_(hy, ny, hx * nx)));
            if (!p.out.normal3) { nx = ny = nz = 0.f; }
        } else {
            px = py = pz = 0.f;
        }
    }
    if (p.out.tile_count) {   // kept rays of this wave's 64 consecutive outputs (feeds lrc_compact_dev)
        const unsigned long long m = __ballot(keep);
        if ((tid & 63u) == 0) p.out.tile_count[gid >> 6] = (uint32_t)__popcll(m);
    }
    if (p.out.t) p.out.t[gid] = t_out;
    if (p.out.t_label) ((uint2*)p.out.t_label)[gid] = make_uint2(__float_as_uint(t_out), label);
    if (p.out.prim) p.out.prim[gid] = prim;
    if (p.out.normal3) { float* q = p.out.normal3 + gid * 3; q[0] = nx; q[1] = ny; q[2] = nz; }
    if (p.out.point3) { float* q = p.out.point3 + gid * 3; q[0] = px; q[1] = py; q[2] = pz; }
    if (p.out.sem) p.out.sem[gid] = (uint16_t)(label & 0xFFFFu);
    if (p.out.ins) p.out.ins[gid] = (uint16_t)(label >> 16);
    if (p.out.incident_deg) p.out.incident_deg[gid] = inc;
    if (p.out.intensity) p.out.intensity[gid] = inten;

}

constexpr int kStatsWords = 5;   // node steps, triangle tests, wave-uniform node steps, dead node steps, pad-clause rejections

// ---- stable compaction, one tile (64 consecutive entries of one segment) per wave ---------------------------------------
// exclusive prefix of the super-tile totals at super tile s, formed on the spot: a wave sums at most a few hundred
// numbers, which is cheaper than the launch of a kernel that would tabulate them (compact_base_kernel)
__device__ __forceinline__ uint64_t super_prefix_wave(const uint32_t* __restrict__ total, uint64_t s, uint32_t lane) {
    uint64_t acc = 0;
    for (uint64_t j = lane; j < s; j += 64) acc += total[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    return acc;
}
__device__ __forceinline__ uint64_t super_prefix_thread(const uint32_t* __restrict__ total, uint64_t s) {
    uint64_t acc = 0;
    for (uint64_t j = 0; j < s; ++j) acc += total[j];
    return acc;
}

// one thread per segment: count = global offset of its end tile - global offset of its first tile
__device__ __forceinline__ void segment_count(const lrc_compact_io& io, uint64_t tps, uint64_t ntiles, uint64_t nseg, uint64_t sg,
                                              const uint32_t* tile_off, const uint64_t* super_base, const uint32_t* super_total) {
    if (sg >= nseg) return;
    const uint64_t nsuper = (ntiles + 1023) / 1024;
    const uint64_t g0 = sg * tps, g1 = g0 + tps;
    auto at = [&](uint64_t g) -> uint64_t {
        if (super_total) return g >= ntiles ? super_prefix_thread(super_total, nsuper)
                                            : super_prefix_thread(super_total, g >> 10) + tile_off[g];
        return g >= ntiles ? super_base[nsuper] : super_base[g >> 10] + tile_off[g];
    };
    io.counts[sg] = at(g1) - at(g0);
}

// the kept entries of tile `tile` go to their rows; called by all 64 lanes of one wave
__device__ __forceinline__ void scatter_tile(const lrc_compact_io& io, uint64_t seg_len, uint64_t tps, uint64_t tile, uint32_t lane,
                                             const uint32_t* tile_off, const uint64_t* super_base, uint64_t tile_base,
                                             const uint32_t* super_total) {
    const uint64_t seg = tile / tps, chunk = tile - seg * tps;
    const uint64_t i = chunk * 64 + lane;
    const uint64_t src = seg * seg_len + i;
    bool keep = false;
    if (i < seg_len) keep = io.t[src] < __builtin_inff();
    const unsigned long long m = __ballot(keep);
    if (m == 0ull) return;
    const uint64_t gt = tile_base + tile;
    const uint64_t sbase = super_total ? super_prefix_wave(super_total, gt >> 10, lane) : super_base[gt >> 10];
    if (!keep) return;
    const uint64_t tbase = sbase + tile_off[gt];
    const uint64_t dst = tbase + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
    if (io.out_xyzl) {
        const float* sp = io.point3 + src * 3;
        const uint32_t lab = (io.sem ? (uint32_t)io.sem[src] : 0u) | ((io.ins ? (uint32_t)io.ins[src] : 0u) << 16);
        ((float4*)io.out_xyzl)[dst] = make_float4(sp[0], sp[1], sp[2], __uint_as_float(lab));
    }
    if (io.out_point3) {
        const float* sp = io.point3 + src * 3;
        float* q = io.out_point3 + dst * 3;
        q[0] = sp[0]; q[1] = sp[1]; q[2] = sp[2];
    }
    if (io.out_sem) io.out_sem[dst] = io.sem[src];
    if (io.out_ins) io.out_ins[dst] = io.ins[src];
    if (io.out_incident_deg) io.out_incident_deg[dst] = io.incident_deg[src];
    if (io.out_index) io.out_index[dst] = (uint32_t)i;
    if (io.out_range_origin) {
        // |p| from the WORLD origin in float32 exactly as np.linalg.norm(points, axis=1) forms it: squares, the
        // 3-element add.reduce left to right, one sqrt -- the quantity the reference's ScanQuality range statistics are
        // taken over (s3dis_simulator.py:283-284)
        const float* sp = io.point3 + src * 3;
        io.out_range_origin[dst] = __builtin_sqrtf((sp[0] * sp[0] + sp[1] * sp[1]) + sp[2] * sp[2]);
    }
}

// R consecutive tiles by one wave, packed (x, y, z, label) rows only, tiles aligned with the segments (seg_len % 64 == 0) and R
// dividing 1024 (the tiles of a call lie in one super tile).  This is the form the trace launch of the scan pipeline runs in
// its leading workgroups: a wave there holds a trace wave's registers and LDS, so few waves with many loads in flight each --
// all keep flags first, then all payloads, then the stores: two memory round trips per R tiles instead of per tile.
// super_base != NULL: the offsets are those of a scan over MORE tiles than this call's (all ranks' scans): tile t of the call is
// tile tile_base + t of that scan and its super tile's base is tabulated (compact_base_kernel); else the bases are summed here
// from the totals and the R tiles lie in one super tile.
template <int R>
__device__ __forceinline__ void scatter_tiles_xyzl(const lrc_compact_io& io, uint64_t ntiles, uint64_t tile0, uint32_t lane,
                                                   const uint32_t* tile_off, const uint32_t* super_total,
                                                   uint64_t tile_base = 0, const uint64_t* super_base = nullptr) {
    float t[R];
    uint64_t off[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t tile = tile0 + r;
        const bool valid = tile < ntiles;
        t[r] = valid ? io.t[tile * 64 + lane] : __builtin_inff();
        off[r] = valid ? (uint64_t)tile_off[tile_base + tile] : 0u;
        if (super_base && valid) off[r] += super_base[(tile_base + tile) >> 10];
    }
    const uint64_t sbase = super_base ? 0ull : super_prefix_wave(super_total, tile0 >> 10, lane);
    float px[R], py[R], pz[R];
    uint32_t lab[R];
    unsigned long long m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool keep = t[r] < __builtin_inff();
        m[r] = __ballot(keep);
        px[r] = py[r] = pz[r] = 0.0f;
        lab[r] = 0u;
        if (keep) {
            const uint64_t src = (tile0 + r) * 64 + lane;
            const float* sp = io.point3 + src * 3;
            px[r] = sp[0]; py[r] = sp[1]; pz[r] = sp[2];
            lab[r] = (io.sem ? (uint32_t)io.sem[src] : 0u) | ((io.ins ? (uint32_t)io.ins[src] : 0u) << 16);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (t[r] < __builtin_inff()) {
            const uint64_t dst = sbase + off[r] + (uint64_t)__popcll(m[r] & ((1ull << lane) - 1ull));
            ((float4*)io.out_xyzl)[dst] = make_float4(px[r], py[r], pz[r], __uint_as_float(lab[r]));
        }
    }
}
#ifndef LRC_PRE_TILES
#define LRC_PRE_TILES 8
#endif
constexpr int kPreTiles = LRC_PRE_TILES;      // tiles per leading workgroup of a pipelined trace launch (4 / 8 / 16 measured)

template <int I> struct IntTag { static constexpr int value = I; };

// GEN: 0 = explicit rays, 1 = pose x direction table, 2 = pose x per-ray scan angles (dual-axis sensor, opt-in)
// QN: 1 = walk the quantised node images (32-byte nodes for the per-lane fetches), 2 = walk their
//     four-wide collapse (64-byte nodes, half the steps); a wave with a ray outside the bound the quantisation margin is
//     proven for walks the float32 world-space nodes instead
template <int GEN, int LEAFW, bool UNI, bool SPEC, bool STATS = false, int QN = 0>
__global__ __launch_bounds__(kTBlock, (QN != 0 && !STATS && GEN == 1) ? 8 : 1) void trace_kernel(const TraceParams p) {
    extern __shared__ int s_stack[];   // [stack depth][kTBlock]: one column per lane, conflict free
    const uint32_t tid = threadIdx.x;
    if (GEN == 1 && p.pre.blocks != 0u && p.pre.sharded != 0u) {
        if (blockIdx.x < p.pre.blocks) {
            // per-pose counts of the assembled cloud (one thread per pose of ALL ranks), then this workgroup's share
            rebuild_counts(p.pre.rq, blockIdx.x * kTBlock + tid, p.pre.blocks * kTBlock);
            if (blockIdx.x < p.pre.own_blocks) {
                const uint64_t tile0 = (uint64_t)blockIdx.x * kPreTiles;
                if (tile0 < p.pre.ntiles)
                    scatter_tiles_xyzl<kPreTiles>(p.pre.io, p.pre.ntiles, tile0, tid, p.pre.tile_off, nullptr, p.pre.tile_base,
                                                  p.pre.super_base);
            } else {
                const uint32_t tile0 = (blockIdx.x - p.pre.own_blocks) * (uint32_t)LRC_REBUILD_R;
                if (tile0 < p.pre.rq.ntiles) rebuild_tiles<LRC_REBUILD_R>(p.pre.rq, tile0, p.pre.rq.ntiles, tid);
            }
            return;
        }
    } else
    if (GEN == 1 && p.pre.blocks != 0u) {          // wave-uniform; only the pose-batched scan is ever pipelined
        if (blockIdx.x < p.pre.blocks) {
            if (p.pre.io.counts) segment_count(p.pre.io, p.pre.tps, p.pre.ntiles, p.pre.nseg, (uint64_t)blockIdx.x * kTBlock + tid,
                                               p.pre.tile_off, nullptr, p.pre.super_total);
            const uint64_t tile0 = (uint64_t)blockIdx.x * kPreTiles;
            if (tile0 < p.pre.ntiles) {
                if (p.pre.rows_only) {
                    scatter_tiles_xyzl<kPreTiles>(p.pre.io, p.pre.ntiles, tile0, tid, p.pre.tile_off, p.pre.super_total);
                } else {
                    for (int r = 0; r < kPreTiles; ++r)
                        if (tile0 + r < p.pre.ntiles)
                            scatter_tile(p.pre.io, p.pre.seg_len, p.pre.tps, tile0 + r, tid, p.pre.tile_off, nullptr, 0, p.pre.super_total);
                }
            }
            return;
        }
    }
    const uint32_t tile = p.tile_chunk_log2 ? xcd_tile_chunked(blockIdx.x - p.pre.blocks, gridDim.x - p.pre.blocks, p.tile_chunk_log2)
                                            : xcd_tile(blockIdx.x - p.pre.blocks, gridDim.x - p.pre.blocks);
    const uint64_t gid = (uint64_t)tile * kTBlock + tid;
    // launch chain: workgroups are handed out in blockIdx order, so when the last one starts the launch has no workgroup
    // left to dispatch -- the next trace launch (held on another stream behind this word) may start filling the freed slots
    if (p.chain_word != nullptr && blockIdx.x == gridDim.x - 1 && tid == 0)
        __hip_atomic_store(p.chain_word, p.chain_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (gid >= p.total) return;

    // ---- the ray ----
    V3 o, d;
    double cx, cy, cz;
    bool live = true;          // false: not cast at all (dropped by the sensor, or a non-finite ray)
    uint32_t pose32 = 0;
    if (GEN == 1) {
        const uint64_t pose = gid / p.rays_per_pose;
        const uint64_t i = gid - pose * p.rays_per_pose;
        gen_ray(p.poses16, p.dirs3, pose, i, o, d, cx, cy, cz);
        pose32 = (uint32_t)pose;
    } else if (GEN == 2) {
        const uint64_t pose = gid / p.rays_per_pose;
        const double2 a = ((const double2*)p.angles2)[gid];
        gen_ray_angles(p.poses16, pose, a.x, a.y, o, d, cx, cy, cz);
        if (p.keep_mask) live = p.keep_mask[gid] != 0;
        pose32 = (uint32_t)pose;
    } else {
        const float* r = p.rays6 + gid * 6;
        o.x = r[0]; o.y = r[1]; o.z = r[2];
        d.x = r[3]; d.y = r[4]; d.z = r[5];
        if (p.keep_mask) live = p.keep_mask[gid] != 0;          // rays the sensor dropped: never cast
        if (!p.seg_offsets && p.seg_centers3) {
            // explicit rays of several poses at a FIXED stride (rays_per_pose each, dropped ones masked): pose = gid / N
            pose32 = (uint32_t)(gid / p.rays_per_pose);
        } else if (p.seg_offsets) {
            // rays of several poses back to back: find this ray's pose (largest s with off[s] <= gid)
            uint32_t lo = 0, hi = p.num_segments;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (p.seg_offsets[mid] <= gid) lo = mid; else hi = mid;
            }
            pose32 = lo;
        }
    }

    // ---- closest hit ----
    float tbest = __builtin_inff();
    uint32_t best_slot = 0xFFFFFFFFu;
    uint32_t best_prim = 0xFFFFFFFFu;   // loaded lazily, only to break exact ties

    uint32_t st_nodes = 0, st_tris = 0, st_uni = 0, st_dead = 0, st_pad = 0;   // STATS build only (lrc_debug_scan_stats)
    live = live & finite_ray(o, d);

    // The traversal, once per node image: Q = false walks the float32 world-space nodes with the ray's world-space slab
    // constants; Q = true walks the quantised images with the slab constants in normalised coordinates.  Which boxes
    // are visited differs (the quantised boxes are a little larger), what is found does not: every box test is
    // conservative with respect to the hit definition (lrc_device.h), the closest hit is order independent.
    // qtag 0: float32 world-space nodes.  1: quantised images; all rays of the wave point into one direction octant, so
    // the plane selectors are wave-uniform and live in SGPRs.  2: the same on the four-wide collapse of the tree.
    bool redo = false;         // quantised paths: this ray must be redone on qtag 0 (its closest candidate failed the box
                               // clause, or -- qtag 2 -- its stack would not have held a step's pushes)
    auto traverse = [&](auto qtag, auto ctag) {
        constexpr int QM = decltype(qtag)::value;
        constexpr bool Q = QM != 0;
        constexpr bool INLINE_CLAUSE = decltype(ctag)::value != 0;    // the redo route tests the clause per test
        (void)INLINE_CLAUSE;
        RaySlab sl;
        uint32_t sel_nx = 0, sel_ny = 0, sel_nz = 0, sel_fx = 0, sel_fy = 0, sel_fz = 0;
        if (Q) {
            // normalised coordinates: n = (x - base) / W per axis, W a power of two (scaling by it is exact)
            const float ix = safe_inv(d.x), iy = safe_inv(d.y), iz = safe_inv(d.z);
            sl.ix = ix * p.qW[0]; sl.iy = iy * p.qW[1]; sl.iz = iz * p.qW[2];
            sl.ox = ((o.x - p.qbase[0]) * p.qinvW[0]) * sl.ix;
            sl.oy = ((o.y - p.qbase[1]) * p.qinvW[1]) * sl.iy;
            sl.oz = ((o.z - p.qbase[2]) * p.qinvW[2]) * sl.iz;
            // v_perm_b32 selectors: the plane the rays enter / leave through on each axis, picked by the sign of the
            // direction (for ix > 0 fma(lo, ix, -ox) <= fma(hi, ix, -ox) by monotone rounding, so picking by sign IS
            // the min / max of slab_interval, bit for bit)
            constexpr uint32_t kLo = 0x0701000Cu, kHi = 0x0703020Cu;   // {0x40, half word, 0x00}
            const uint32_t oct = (uint32_t)__builtin_amdgcn_readfirstlane((int)sign_octant(d));
            sel_nx = (oct & 1u) ? kHi : kLo; sel_fx = (oct & 1u) ? kLo : kHi;
            sel_ny = (oct & 2u) ? kHi : kLo; sel_fy = (oct & 2u) ? kLo : kHi;
            sel_nz = (oct & 4u) ? kHi : kLo; sel_fz = (oct & 4u) ? kLo : kHi;
        } else {
            sl = make_slab(o, d);
        }
        int sp = 0;
        int ref = 0;   // root
        // descend into the nearer hit child, push the other; nothing hit -> pop, or (stack empty) continue with the
        // empty leaf so that the outer loop ends
        auto choose = [&](float n0, float f0, float n1, float f1, int r0, int r1) {
            if (STATS) st_nodes += 1u;
            const bool h0 = (n0 <= f0) & (n0 <= tbest);
            const bool h1 = (n1 <= f1) & (n1 <= tbest);
            if (h0 & h1) {
                const bool first0 = n0 <= n1;
                s_stack[sp * kTBlock + tid] = first0 ? r1 : r0;
                ++sp;
                ref = first0 ? r0 : r1;
            } else if (h0) {
                ref = r0;
            } else if (h1) {
                ref = r1;
            } else if (sp == 0) {
                if (STATS) st_dead += 1u;
                ref = ~0;   // empty leaf
            } else {
                if (STATS) st_dead += 1u;
                --sp;
                ref = s_stack[sp * kTBlock + tid];
            }
        };
        // one inner-node step on a float32 node (world space, or the normalised image on the scalar path)
        auto step = [&](const F4 q0, const F4 q1, const F4 q2, const F4 q3) {
            float n0, f0, n1, f1;
            slab_interval(sl, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, n0, f0);
            slab_interval(sl, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, n1, f1);
            choose(n0, f0, n1, f1, __float_as_int(q3.x), __float_as_int(q3.y));
        };
        // one inner-node step on a 32-byte quantised node: per child (lo|hi << 16) x, y, z and the reference
        auto step_q = [&](const uint4 a, const uint4 b) {
            auto plane = [](uint32_t w, uint32_t sel) { return __uint_as_float(__builtin_amdgcn_perm(0x40000000u, w, sel)); };
            auto interval = [&](const uint4 c, float& tn, float& tf) {
                const float nx = fma_(plane(c.x, sel_nx), sl.ix, -sl.ox), fx = fma_(plane(c.x, sel_fx), sl.ix, -sl.ox);
                const float ny = fma_(plane(c.y, sel_ny), sl.iy, -sl.oy), fy = fma_(plane(c.y, sel_fy), sl.iy, -sl.oy);
                const float nz = fma_(plane(c.z, sel_nz), sl.iz, -sl.oz), fz = fma_(plane(c.z, sel_fz), sl.iz, -sl.oz);
                const float n = max2(max2(nx, ny), max2(nz, 0.0f));
                const float f = min2(min2(fx, fy), fz);
                tn = fma_(n, kPadRelLo, -kPadAbs);
                tf = fma_(f, kPadRelHi, kPadAbs);
            };
            float n0, f0, n1, f1;
            interval(a, n0, f0);
            interval(b, n1, f1);
            choose(n0, f0, n1, f1, (int)a.w, (int)b.w);
        };
#ifdef LRC_VARIANTS
        // four-wide step: descend into the nearest hit child, push the other hit children; nothing hit -> pop
        auto choose4 = [&](float t0, float f0, float t1, float f1, float t2, float f2, float t3, float f3,
                           int r0, int r1, int r2, int r3) {
            if (STATS) st_nodes += 1u;
            const bool h0 = (t0 <= f0) & (t0 <= tbest), h1 = (t1 <= f1) & (t1 <= tbest);
            const bool h2 = (t2 <= f2) & (t2 <= tbest), h3 = (t3 <= f3) & (t3 <= tbest);
            const float inf = __builtin_inff();
            const float d0 = h0 ? t0 : inf, d1 = h1 ? t1 : inf, d2 = h2 ? t2 : inf, d3 = h3 ? t3 : inf;
            const float dm = min2(min2(d0, d1), min2(d2, d3));
            if (dm < inf) {
                if (sp + 3 > (int)p.stack_cap) {      // cannot happen on a tree whose depth the stack was sized for
                    redo = true; ref = ~0; sp = 0;    // ... unless three children stay pending level after level
                } else {
                    const int near = d0 == dm ? 0 : d1 == dm ? 1 : d2 == dm ? 2 : 3;
                    if (h3 & (near != 3)) { s_stack[sp * kTBlock + tid] = r3; ++sp; }
                    if (h2 & (near != 2)) { s_stack[sp * kTBlock + tid] = r2; ++sp; }
                    if (h1 & (near != 1)) { s_stack[sp * kTBlock + tid] = r1; ++sp; }
                    if (h0 & (near != 0)) { s_stack[sp * kTBlock + tid] = r0; ++sp; }
                    ref = near == 0 ? r0 : near == 1 ? r1 : near == 2 ? r2 : r3;
                }
            } else if (sp == 0) {
                if (STATS) st_dead += 1u;
                ref = ~0;
            } else {
                if (STATS) st_dead += 1u;
                --sp;
                ref = s_stack[sp * kTBlock + tid];
            }
        };
        auto step_q4 = [&](const uint4 a, const uint4 b, const uint4 c, const uint4 e) {
            auto plane = [](uint32_t w, uint32_t sel) { return __uint_as_float(__builtin_amdgcn_perm(0x40000000u, w, sel)); };
            auto interval = [&](const uint4 k, float& tn, float& tf) {
                const float nx = fma_(plane(k.x, sel_nx), sl.ix, -sl.ox), fx = fma_(plane(k.x, sel_fx), sl.ix, -sl.ox);
                const float ny = fma_(plane(k.y, sel_ny), sl.iy, -sl.oy), fy = fma_(plane(k.y, sel_fy), sl.iy, -sl.oy);
                const float nz = fma_(plane(k.z, sel_nz), sl.iz, -sl.oz), fz = fma_(plane(k.z, sel_fz), sl.iz, -sl.oz);
                const float n = max2(max2(nx, ny), max2(nz, 0.0f));
                const float f = min2(min2(fx, fy), fz);
                tn = fma_(n, kPadRelLo, -kPadAbs);
                tf = fma_(f, kPadRelHi, kPadAbs);
            };
            float t0, f0, t1, f1, t2, f2, t3, f3;
            interval(a, t0, f0); interval(b, t1, f1); interval(c, t2, f2); interval(e, t3, f3);
            choose4(t0, f0, t1, f1, t2, f2, t3, f3, (int)a.w, (int)b.w, (int)c.w, (int)e.w);
        };
        // the same on a wave-uniform node of the normalised float32 image (128 B: per child lo, hi, reference, pad)
        auto step_n4 = [&](const F4 a0, const F4 a1, const F4 b0, const F4 b1, const F4 c0, const F4 c1, const F4 e0,
                           const F4 e1) {
            float t0, f0, t1, f1, t2, f2, t3, f3;
            slab_interval(sl, a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, t0, f0);
            slab_interval(sl, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, t1, f1);
            slab_interval(sl, c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, t2, f2);
            slab_interval(sl, e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, t3, f3);
            choose4(t0, f0, t1, f1, t2, f2, t3, f3, __float_as_int(a1.z), __float_as_int(b1.z), __float_as_int(c1.z),
                    __float_as_int(e1.z));
        };
#endif
        // one leaf: test its 1..4 triangles, keep the lexicographically smallest (t, triangle row)
        auto leaf = [&](const int lref) {
            const uint32_t enc = (uint32_t)(~lref);
            const uint32_t first = enc >> 3, cnt = enc & 7u;
            // LEAFW records are fetched at once (one memory round trip per LEAFW triangles);
            // the triangle array is padded so that reading past a short leaf stays in bounds
            for (uint32_t k0 = 0; k0 < cnt; k0 += LEAFW) {
                float4 ra[LEAFW], rb[LEAFW], rc[LEAFW];
#pragma unroll
                for (int j = 0; j < LEAFW; ++j) {
                    const float4* tr = p.tris + (size_t)(first + k0 + j) * 3;
#ifdef LRC_TRI36
                    // A/B only (profiles/r04_trace_levers.txt): 36 of the record's 48 bytes come back, Ng = cross(e2, e1) is
                    // formed again (the builder's own expression: same bits) -- one dwordx4 return per test for six VALU
                    ra[j] = tr[0]; rb[j] = tr[1]; rc[j].x = ((const float*)tr)[8];
#else
                    ra[j] = tr[0]; rb[j] = tr[1]; rc[j] = tr[2];
#endif
                }
#pragma unroll
                for (int j = 0; j < LEAFW; ++j) {
                    if (k0 + j < cnt) {
                        if (STATS) st_tris += 1u;
                        const uint32_t slot = first + k0 + j;
                        const float4 a = ra[j], b = rb[j], c = rc[j];
#ifdef LRC_TRI36
                        const V3 v0{a.x, a.y, a.z}, v1{a.w, b.x, b.y}, v2{b.z, b.w, c.x}, ng = cross3(v2, v1);
#else
                        const V3 v0{a.x, a.y, a.z}, v1{a.w, b.x, b.y}, v2{b.z, b.w, c.x}, ng{c.y, c.z, c.w};
#endif
                        float t;
                        bool hit;
                        // edge records (v0, e1, e2, Ng): v1 / v2 above ARE e1 / e2.  Candidates are ranked by the
                        // Moeller-Trumbore conditions on every node image; the box clause is tested once, after the
                        // traversal (below), and inline only on the redo route (INLINE_CLAUSE, float32 nodes).
                        hit = tri_mt(o, d, v0, v1, v2, ng, t);
                        if (INLINE_CLAUSE) {
                            if (hit) {
                                const float* bx = p.slot_box + (size_t)slot * 6;
                                hit = box_clause(sl, bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], t);
                                if (STATS) { if (!hit) st_pad += 1u; }
                            }
                        }
                        if (hit) {
                            if (t < tbest) {
                                tbest = t; best_slot = slot; best_prim = 0xFFFFFFFFu;
                            } else if (t == tbest) {
                                if (best_prim == 0xFFFFFFFFu) best_prim = p.slot_prim[best_slot];
                                const uint32_t pr = p.slot_prim[slot];
                                if (pr < best_prim) { best_slot = slot; best_prim = pr; }
                            }
                        }
                    }
                }
            }
        };
        auto fetch_step = [&]() {         // per-lane fetch of node `ref`
#ifdef LRC_VARIANTS
            if (QM == 2) {
                const uint4* n = p.nodes_q4 + (size_t)ref * 4;
                step_q4(n[0], n[1], n[2], n[3]);
            } else
#endif
            if (Q) {
                const uint4* n = p.nodes_q + (size_t)ref * 2;
                step_q(n[0], n[1]);
            } else {
                const F4* n = (const F4*)(p.nodes + (size_t)ref * 4);
                step(n[0], n[1], n[2], n[3]);
            }
        };
        while (true) {
            // descend inner nodes
            int pending = ~0;   // SPEC: one postponed leaf (empty = ~0)
            while (ref >= 0) {
                if (UNI) {
                    // Neighbouring rays walk the top of the tree in lock step.  When every active lane
                    // of the wave wants the same node, fetch it once through the scalar cache into
                    // SGPRs (s_load) instead of 64 identical vector loads through the L1.
                    const int uref = __builtin_amdgcn_readfirstlane(ref);
                    if (__builtin_amdgcn_ballot_w64(ref != uref) == 0ull) {
                        if (STATS) st_uni += 1u;
#ifdef LRC_VARIANTS
                        if (QM == 2) {
                            const float4* n = p.nodes_n4 + (size_t)uref * 8;
                            step_n4(ld_uniform(n), ld_uniform(n + 1), ld_uniform(n + 2), ld_uniform(n + 3),
                                    ld_uniform(n + 4), ld_uniform(n + 5), ld_uniform(n + 6), ld_uniform(n + 7));
                        } else
#endif
                        {
                            const float4* n = (Q ? p.nodes_n : p.nodes) + (size_t)uref * 4;
                            step(ld_uniform(n), ld_uniform(n + 1), ld_uniform(n + 2), ld_uniform(n + 3));
                        }
                    } else {
                        fetch_step();
                    }
                } else {
                    fetch_step();
                }
                if (SPEC) {
                    // speculative traversal: park the first leaf and keep descending from the stack, so
                    // that this lane stays busy while its neighbours are still in inner nodes
                    if ((ref < 0) & (ref != ~0) & (pending == ~0) & (sp > 0)) {
                        pending = ref;
                        --sp;
                        ref = s_stack[sp * kTBlock + tid];
                    }
                }
            }
            if (SPEC) leaf(pending);
            leaf(ref);
            if (sp == 0) break;
            --sp;
            ref = s_stack[sp * kTBlock + tid];
        }
        if (!INLINE_CLAUSE) {
            if (best_slot != 0xFFFFFFFFu) {       // the box clause of the closest candidate, in world coordinates
                const float* bx = p.slot_box + (size_t)best_slot * 6;
                RaySlab w = sl;
                if (Q) {        // ix = ix' / W exactly (W a power of two); ox = o * ix as make_slab forms it
                    w.ix = sl.ix * p.qinvW[0]; w.iy = sl.iy * p.qinvW[1]; w.iz = sl.iz * p.qinvW[2];
                    w.ox = o.x * w.ix; w.oy = o.y * w.iy; w.oz = o.z * w.iz;
                }
                if (!box_clause(w, bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], tbest)) {
                    redo = true;
                    if (STATS) st_pad += 1u;
                }
            }
#ifdef LRC_VARIANTS
            if (p.force_redo) redo |= (((uint64_t)blockIdx.x * kTBlock + tid) % p.force_redo) == 0;   // test hook (LRC_DEBUG_FORCE_REDO=m)
#endif
        }
    };
    using FirstPass = IntTag<0>;      // edge records: every first pass defers the box clause
    if (p.num_nodes) {
        if (QN) {
            // outside this bound the margin of the quantised boxes is not proven to cover the difference
            // between the two slab arithmetics: such a wave walks the float32 nodes
            auto far1 = [](float oa, float da, float base, float W) {
                return !(__builtin_fabsf(oa - base) <= lrc::kQnodeNearBase * W) | !(__builtin_fabsf(oa) <= lrc::kQnodeNearOrigin * W) |
                       !(__builtin_fabsf(da) <= 0x1p60f);
            };
            const bool far = live & (far1(o.x, d.x, p.qbase[0], p.qW[0]) | far1(o.y, d.y, p.qbase[1], p.qW[1]) |
                                     far1(o.z, d.z, p.qbase[2], p.qW[2]));
            const uint32_t oct = sign_octant(d);
            // octant of the first ACTIVE lane (which may be a ray that is not cast: the wave then merely takes the float32
            // nodes; results do not depend on the choice)
            const uint32_t oct0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)oct);
            // a wave whose rays point into different octants (it straddles an axis direction) walks the float32 nodes too
            if (__builtin_amdgcn_ballot_w64(far | (live & (oct != oct0))) != 0ull) { if (live) traverse(IntTag<0>{}, FirstPass{}); }
            else if (live) {
#ifdef LRC_VARIANTS
                if (QN == 2) traverse(IntTag<2>{}, IntTag<0>{});
                else
#endif
                traverse(IntTag<1>{}, IntTag<0>{});
            }
        } else if (live) {
            traverse(IntTag<0>{}, FirstPass{});
        }
        // a ray whose closest candidate failed the box clause (or, four-wide nodes, whose stack was too short) is redone
        // on the float32 nodes with the clause tested per triangle
        if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
            if (redo) {
                tbest = __builtin_inff(); best_slot = 0xFFFFFFFFu; best_prim = 0xFFFFFFFFu;
                traverse(IntTag<0>{}, IntTag<1>{});
            }
        }
    }

    // ---- fused write-back ----
    if (GEN != 0) {
        // the range-filter centre (the pose's translation, float64) is fetched again here instead of being held in six
        // registers through the traversal; the pointer is made opaque so that the fetch is not merged with gen_ray's
        const double* M = p.poses16;
        asm volatile("" : "+s"(M));
        M += (size_t)pose32 * 16;
        cx = M[3]; cy = M[7]; cz = M[11];
    } else if (p.seg_centers3) {
        const double* c = p.seg_centers3 + (size_t)pose32 * 3;      // the centre of this ray's pose (segment)
        cx = c[0]; cy = c[1]; cz = c[2];
    } else if (p.has_center) { cx = p.cx; cy = p.cy; cz = p.cz; }
    else { cx = (double)o.x; cy = (double)o.y; cz = (double)o.z; }
    // the ray's index is formed again from the workgroup number (scalar arithmetic) instead of being carried through the
    // traversal in two registers the kernel does not have (the compiler spilled them to scratch: 8 B per ray each way);
    // the workgroup number is made opaque so that the two computations are not merged
    uint32_t wg = blockIdx.x;
    asm volatile("" : "+s"(wg));
    const uint32_t tile_w = p.tile_chunk_log2 ? xcd_tile_chunked(wg - p.pre.blocks, gridDim.x - p.pre.blocks, p.tile_chunk_log2)
                                              : xcd_tile(wg - p.pre.blocks, gridDim.x - p.pre.blocks);
    const uint64_t gid_w = (uint64_t)tile_w * kTBlock + tid;
    write_back<GEN != 0>(p, gid_w, tid, o, d, cx, cy, cz, tbest, best_slot);
    if (STATS) {
        if (p.stats) {
            uint32_t* q = p.stats + gid_w * kStatsWords;
            q[0] = st_nodes; q[1] = st_tris; q[2] = st_uni; q[3] = st_dead; q[4] = st_pad;
        }
    }

}

#ifdef LRC_VARIANTS
// ---- measured alternative: K rays per lane with private refill (LRC_REFILL=K) -------------
// One wave owns K consecutive 64-ray tiles of a pose-batched scan; lane l traces rays l, 64+l, 128+l, ... one after the
// other, starting its next ray the moment the current one is done instead of idling until the slowest lane of the wave
// has finished ("private refill").  The closest hit (t, slot) of each ray is parked in LDS and the whole write-back
// runs afterwards, coherently, K passes of 64 lanes.  The next ray's direction is fetched while the current one is
// traced.  Same arithmetic, same result bytes as trace_kernel (tests/test_parity_gpu.py::
// test_kernel_variants_are_bit_identical); what changes is the balance of work inside a wave (tools/trav_stats.py:
// 0.68 -> 0.77 / 0.84 for K = 2 / 4) at the price of (a) refilled lanes restarting at the root while their
// neighbours are deep in the tree, which takes the wave off the scalar-fetch path, (b) K x 512 B more LDS per wave and
// 6 more VGPRs, i.e. fewer resident waves.
// W = resident waves per SIMD the register allocator must leave room for (its natural footprint is 78 / 93 VGPRs for
// K = 2 / 4, i.e. 6 / 5 waves; 7 waves cost a few spilled registers)
template <int K, int W>
__global__ __launch_bounds__(kTBlock, W) void trace_refill_kernel(const TraceParams p, uint32_t depth) {
    extern __shared__ int s_mem[];                 // [depth][64] stacks | [K][64] {t bits, slot}
    int* s_stack = s_mem;
    uint2* s_res = (uint2*)(s_mem + (size_t)depth * 64);
    const uint32_t tid = threadIdx.x;
    // the host launches this kernel only when rays_per_pose % (64 K) == 0: a wave's K tiles lie in ONE pose, so the pose
    // and everything derived from it is wave-uniform (scalar loads, SGPR operands)
    const uint64_t N = p.rays_per_pose;
    const uint64_t tile0 = (uint64_t)xcd_tile(blockIdx.x, gridDim.x) * (64u * K);
    if (tile0 >= p.total) return;
    const uint64_t pose = tile0 / N;
    const uint32_t i0 = (uint32_t)(tile0 - pose * N);
    const double* M = p.poses16 + pose * 16;
    const uint64_t base = tile0 + tid;

    V3 o, d;
    RaySlab sl;
    float tbest;
    uint32_t best_slot, best_prim;
    int sp = 0, ref = ~0;
    int k = 0;
    double nd0 = 0.0, nd1 = 0.0, nd2 = 0.0;        // next ray's sensor-frame direction, in flight

    auto fetch_next = [&](int kk) {
        if (kk < K) {
            const double* dv = p.dirs3 + (size_t)(i0 + (uint32_t)kk * 64u + tid) * 3;
            nd0 = dv[0]; nd1 = dv[1]; nd2 = dv[2];
        }
    };
    // ray kk from the prefetched direction; false when the lane has no ray kk
    auto begin = [&](int kk) -> bool {
        if (kk >= K) return false;
        d.x = (float)dgemm_row(nd0, nd1, nd2, M[0], M[1], M[2]);
        d.y = (float)dgemm_row(nd0, nd1, nd2, M[4], M[5], M[6]);
        d.z = (float)dgemm_row(nd0, nd1, nd2, M[8], M[9], M[10]);
        o.x = (float)M[3]; o.y = (float)M[7]; o.z = (float)M[11];
        sl = make_slab(o, d);
        tbest = __builtin_inff();
        best_slot = 0xFFFFFFFFu; best_prim = 0xFFFFFFFFu;
        sp = 0;
        ref = (p.num_nodes && finite_ray(o, d)) ? 0 : ~0;
        return true;
    };
    fetch_next(0);
    begin(0);
    fetch_next(1);

    auto step = [&](const F4 q0, const F4 q1, const F4 q2, const F4 q3) {
        float n0, f0, n1, f1;
        slab_interval(sl, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, n0, f0);
        slab_interval(sl, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, n1, f1);
        const bool h0 = (n0 <= f0) & (n0 <= tbest);
        const bool h1 = (n1 <= f1) & (n1 <= tbest);
        const int r0 = __float_as_int(q3.x), r1 = __float_as_int(q3.y);
        if (h0 & h1) {
            const bool first0 = n0 <= n1;
            s_stack[sp * 64 + tid] = first0 ? r1 : r0;
            ++sp;
            ref = first0 ? r0 : r1;
        } else if (h0) {
            ref = r0;
        } else if (h1) {
            ref = r1;
        } else if (sp == 0) {
            ref = ~0;
        } else {
            --sp;
            ref = s_stack[sp * 64 + tid];
        }
    };
    auto leaf = [&](const int lref) {
        const uint32_t enc = (uint32_t)(~lref);
        const uint32_t first = enc >> 3, cnt = enc & 7u;
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t slot = first + j;
            const float4* tr = p.tris + (size_t)slot * 3;
            const float4 a = tr[0], b = tr[1], c = tr[2];
            const V3 v0{a.x, a.y, a.z}, e1{a.w, b.x, b.y}, e2{b.z, b.w, c.x}, ng{c.y, c.z, c.w};
            float t;
            if (tri_hit(o, d, sl, v0, e1, e2, ng, p.slot_box + (size_t)slot * 6, t)) {
                if (t < tbest) {
                    tbest = t; best_slot = slot; best_prim = 0xFFFFFFFFu;
                } else if (t == tbest) {
                    if (best_prim == 0xFFFFFFFFu) best_prim = p.slot_prim[best_slot];
                    const uint32_t pr = p.slot_prim[slot];
                    if (pr < best_prim) { best_slot = slot; best_prim = pr; }
                }
            }
        }
    };

    while (true) {
        while (ref >= 0) {
            const int uref = __builtin_amdgcn_readfirstlane(ref);
            if (__builtin_amdgcn_ballot_w64(ref != uref) == 0ull) {
                const float4* n = p.nodes + (size_t)uref * 4;
                step(ld_uniform(n), ld_uniform(n + 1), ld_uniform(n + 2), ld_uniform(n + 3));
            } else {
                const F4* n = (const F4*)(p.nodes + (size_t)ref * 4);
                step(n[0], n[1], n[2], n[3]);
            }
        }
        leaf(ref);
        if (sp != 0) {
            --sp;
            ref = s_stack[sp * 64 + tid];
            continue;
        }
        // this lane's ray is done: park its hit, start the next one
        s_res[k * 64 + tid] = make_uint2(__float_as_uint(tbest), best_slot);
        ++k;
        if (!begin(k)) break;
        fetch_next(k + 1);
    }

    // ---- coherent write-back, one pass per tile ----
#pragma unroll 1
    for (int kk = 0; kk < K; ++kk) {
        const uint64_t gid = base + (uint64_t)kk * 64u;
        double cx, cy, cz;
        gen_ray(p.poses16, p.dirs3, pose, i0 + (uint32_t)kk * 64u + tid, o, d, cx, cy, cz);
        const uint2 r = s_res[kk * 64 + tid];
        write_back<true>(p, gid, tid, o, d, cx, cy, cz, __uint_as_float(r.x), r.y);
    }
}

#include "lrc_sector.h"
#endif   // LRC_VARIANTS
#include "lrc_stats.h"

// ---- compaction -------------------------------------------------------------------------------
// tile = 64 consecutive entries of one segment = one wave; tile index = seg * tps + chunk.

__global__ __launch_bounds__(kBlock) void compact_count_kernel(const float* t, uint64_t seg_len,
                                                               uint64_t tps, uint64_t ntiles,
                                                               uint32_t* tile_cnt) {
    const uint64_t tile = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const uint64_t seg = tile / tps, chunk = tile - seg * tps;
    const uint64_t i = chunk * 64 + (threadIdx.x & 63u);
    bool keep = false;
    if (i < seg_len) keep = t[seg * seg_len + i] < __builtin_inff();
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63u) == 0) tile_cnt[tile] = (uint32_t)__popcll(m);
}

// Pass A: one WAVE scans a run of 1024 tile counts (a "super tile" = 65536 entries):
// tile_off[tile] = exclusive offset inside the super tile, super_total[b] = its sum.
// The counts may sit in slabs (the gathered send buffers of several ranks): tiles_per_slab consecutive tiles, then the
// next slab slab_stride words further on; a plain array is one slab.
// One-wave workgroups on purpose: this kernel runs beside the trace launch of the caller's other stream, which takes every
// wave slot the moment it is freed.  As a 1024-thread workgroup (16 waves that must start together on ONE CU) the scan
// was not scheduled until that launch had no workgroup left -- 150-170 us for a 5 us kernel, and with it the whole chain
// trace -> scan -> scatter -> next trace fell into phase with the other stream's (profiles/r04_chain_timeline.txt).
constexpr int kScanRows = 16;           // 64 lanes x 16 rows = 1024 tiles per wave and super tile
__global__ __launch_bounds__(64) void compact_scan_kernel(const uint32_t* tile_cnt, uint64_t tiles_per_slab,
                                                          uint64_t slab_stride, uint32_t* tile_off,
                                                          uint64_t ntiles, uint32_t* super_total) {
    const uint32_t lane = threadIdx.x;
    const uint64_t nsuper = (ntiles + 1023) / 1024;
    const bool one_slab = tiles_per_slab >= ntiles;          // a plain array: no slab arithmetic (a 64-bit division per entry)
    // grid-stride over the super tiles: the scan pipeline launches FEW waves on purpose (each must find a free wave slot
    // beside a running trace launch, and they arrive at a trickle), stand-alone callers one wave per super tile
    for (uint64_t sup = blockIdx.x; sup < nsuper; sup += gridDim.x) {
        const uint64_t first = sup * 1024;
        uint32_t c[kScanRows];
#pragma unroll
        for (int r = 0; r < kScanRows; ++r) {          // all loads first (coalesced rows of 64), then the arithmetic
            const uint64_t tile = first + (uint64_t)r * 64 + lane;
            c[r] = 0u;
            if (tile < ntiles) {
                uint64_t idx = tile;
                if (!one_slab) {
                    const uint64_t sb = tile / tiles_per_slab;
                    idx = sb * slab_stride + (tile - sb * tiles_per_slab);
                }
                c[r] = tile_cnt[idx];
            }
        }
        uint32_t carry = 0u;
#pragma unroll
        for (int r = 0; r < kScanRows; ++r) {
            uint32_t incl = c[r];                     // inclusive scan of the row inside the wave
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d, 64);
                if (lane >= (uint32_t)d) incl += up;
            }
            const uint64_t tile = first + (uint64_t)r * 64 + lane;
            if (tile < ntiles) tile_off[tile] = carry + incl - c[r];
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) super_total[sup] = carry;
    }
}

// Pass A2: one wave turns the super-tile totals into exclusive bases (base[nsuper] = grand total).  One wave for the same
// reason as above: a 1024-thread workgroup waits for sixteen free wave slots on one CU while a trace launch is running.
__global__ __launch_bounds__(64) void compact_base_kernel(const uint32_t* super_total, uint64_t* super_base,
                                                          uint64_t nsuper) {
    const uint32_t lane = threadIdx.x;
    uint64_t carry = 0;
    for (uint64_t base = 0; base < nsuper; base += 64) {
        const uint64_t i = base + lane;
        const uint64_t v = i < nsuper ? super_total[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t up = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (i < nsuper) super_base[i] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) super_base[nsuper] = carry;
}

// Pass B: scatter the kept entries; the first workgroups also write the per-segment counts (scatter_tile / segment_count,
// defined in front of the trace kernel, which runs them too: TraceParams::pre).
// tile_base: index of this call's tile 0 in tile_off / super_base (non-zero when the offsets come from a scan over the
// tiles of several ranks and this call scatters one rank's records into the assembled cloud).
// super_total != NULL: the bases of the super tiles are summed here from the totals (super_base is not read).
__global__ __launch_bounds__(kBlock) void compact_scatter_kernel(const lrc_compact_io io,
                                                                 uint64_t seg_len, uint64_t tps,
                                                                 uint64_t ntiles, uint64_t nseg,
                                                                 const uint32_t* tile_off,
                                                                 const uint64_t* super_base, uint64_t tile_base,
                                                                 const uint32_t* super_total) {
    if (io.counts) segment_count(io, tps, ntiles, nseg, (uint64_t)blockIdx.x * kBlock + threadIdx.x, tile_off, super_base, super_total);
    const uint64_t tile = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    scatter_tile(io, seg_len, tps, tile, threadIdx.x & 63u, tile_off, super_base, tile_base, super_total);
}

// |p| of assembled (x, y, z, label) rows from the WORLD origin, float32, as np.linalg.norm(points, axis=1) forms it
// (the quantity the reference's ScanQuality range statistics are taken over, s3dis_simulator.py:283-284)
__global__ __launch_bounds__(kBlock) void rows_range_kernel(const float4* __restrict__ rows, uint64_t n, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 r = rows[i];
    out[i] = __builtin_sqrtf((r.x * r.x + r.y * r.y) + r.z * r.z);
}

// ---- scene cloud from per-ray (t, label) pairs -------------------------------------------------------
// A pose-batched scan is a pure function of (pose, direction table), so a rank that holds the poses and the
// table can rebuild any other rank's hit points from the 8-byte pair alone: this is what the multi-GPU
// all-gather moves instead of 16-byte rows.  Same arithmetic as the trace epilogue (gen_ray + hit_point).

__global__ __launch_bounds__(kBlock) void cloud_count_kernel(const uint2* tl, uint64_t seg_len, uint64_t tps,
                                                             uint64_t ntiles, uint32_t* tile_cnt) {
    const uint64_t tile = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const uint64_t seg = tile / tps, chunk = tile - seg * tps;
    const uint64_t i = chunk * 64 + (threadIdx.x & 63u);
    bool keep = false;
    if (i < seg_len) keep = __uint_as_float(tl[seg * seg_len + i].x) < __builtin_inff();
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63u) == 0) tile_cnt[tile] = (uint32_t)__popcll(m);
}

__global__ __launch_bounds__(kBlock) void cloud_scatter_kernel(const double* poses16, const double* dirs3,
                                                               const uint2* tl, uint64_t seg_len, uint64_t tps,
                                                               uint64_t ntiles, uint64_t nseg,
                                                               const uint32_t* tile_off, const uint64_t* super_base,
                                                               float4* out_xyzl, uint64_t* counts) {
    const uint32_t lane = threadIdx.x & 63u;
    if (counts) {
        const uint64_t sg = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
        if (sg < nseg) {
            const uint64_t nsuper = (ntiles + 1023) / 1024;
            const uint64_t g0 = sg * tps, g1 = g0 + tps;
            const uint64_t o0 = g0 >= ntiles ? super_base[nsuper] : super_base[g0 >> 10] + tile_off[g0];
            const uint64_t o1 = g1 >= ntiles ? super_base[nsuper] : super_base[g1 >> 10] + tile_off[g1];
            counts[sg] = o1 - o0;
        }
    }
    const uint64_t tile = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const uint64_t seg = tile / tps, chunk = tile - seg * tps;
    const uint64_t i = chunk * 64 + lane;
    uint2 rec = make_uint2(0x7F800000u, 0u);
    if (i < seg_len) rec = tl[seg * seg_len + i];
    const float t = __uint_as_float(rec.x);
    const bool keep = t < __builtin_inff();
    const unsigned long long m = __ballot(keep);
    if (!keep) return;
    const uint64_t dst = super_base[tile >> 10] + tile_off[tile] + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
    V3 o, d, h, pt;
    double cx, cy, cz;
    gen_ray(poses16, dirs3, seg, i, o, d, cx, cy, cz);
    hit_point(o, d, t, h, pt);
    out_xyzl[dst] = make_float4(pt.x, pt.y, pt.z, __uint_as_float(rec.y));
}

// ---- scene cloud from per-ray triangle ids ---------------------------------------------------------------
// The hit of a pose-batched scan is a pure function of (pose, direction, triangle): a rank that holds the scene,
// the poses and the direction table rebuilds t with the scan's own ray/triangle test (tri_hit, bit-identical) and
// the point and label from it.  So the multi-GPU all-gather moves the 4-byte primitive id per ray -- Open3D's
// primitive_ids -- and nothing else.  Rays the sender dropped (miss, range filter) carry LRC_INVALID_PRIM.
// Entry (pose p, ray i) lives in slab p / pps at word (p % pps) * seg_len + i; slabs are `stride` words apart.

__global__ __launch_bounds__(kBlock) void prim_count_kernel(const uint32_t* prims, uint64_t pps, uint64_t stride,
                                                            uint64_t seg_len, uint64_t tps, uint64_t ntiles,
                                                            uint32_t* tile_cnt) {
    const uint64_t tile = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const uint64_t seg = tile / tps, chunk = tile - seg * tps;
    const uint64_t i = chunk * 64 + (threadIdx.x & 63u);
    const uint64_t sb = seg / pps;
    bool keep = false;
    if (i < seg_len) keep = prims[sb * stride + (seg - sb * pps) * seg_len + i] != LRC_INVALID_PRIM;
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63u) == 0) tile_cnt[tile] = (uint32_t)__popcll(m);
}

// (N,3) -> x[N] y[N] z[N]: the rebuild reads one direction per lane, and a 24-byte stride costs the vector L1 three
// times the transactions of three unit-stride reads
__global__ __launch_bounds__(kBlock) void dirs_transpose_kernel(const double* __restrict__ dirs3, uint32_t n,
                                                                double* __restrict__ soa) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    soa[i] = dirs3[(size_t)i * 3];
    soa[(size_t)n + i] = dirs3[(size_t)i * 3 + 1];
    soa[2 * (size_t)n + i] = dirs3[(size_t)i * 3 + 2];
}

// Stand-alone rebuild: one wave per R consecutive tiles.
// No XCD-contiguous remap of the tiles here: eight write streams a power of two apart land on the same HBM channels
// in lock step -- measured 540 us against 330 us for round-robin workgroups.
template <int R>
__global__ __launch_bounds__(kBlock) void prim_scatter_kernel(const RebuildParams q) {
    rebuild_counts(q, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock);
    // everything that depends on the tile alone is wave-uniform: keep it on the scalar unit
    const uint32_t tile0 = __builtin_amdgcn_readfirstlane((blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * R);
    if (tile0 >= q.ntiles) return;
    rebuild_tiles<R>(q, tile0, q.ntiles, threadIdx.x & 63u);
}

#ifdef LRC_VARIANTS
// Bounding sphere of every triangle's axis-aligned box (centre and half diagonal, rounded up): what sector_kernel
// tests against a packet of rays before it runs the exact ray/triangle test.  Built on first use of the packet kernel.
__global__ __launch_bounds__(kBlock) void slot_sphere_kernel(const float* __restrict__ slot_box, uint32_t num_slots,
                                                             float4* __restrict__ sphere) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= num_slots) return;
    const float* bx = slot_box + (size_t)k * 6;
    float ctr[3];
    double hd2 = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {       // distance from the ROUNDED centre to the farthest corner
        const double lo = (double)bx[q], hi = (double)bx[3 + q];
        ctr[q] = (float)(0.5 * (lo + hi));
        const double cc = (double)ctr[q];
        const double e = fmax(hi - cc, cc - lo);
        hd2 += e * e;
    }
    const float r = (float)__builtin_sqrt(hd2);
    sphere[k] = make_float4(ctr[0], ctr[1], ctr[2], __uint_as_float(__float_as_uint(r) + 1u));    // nextafter(r, +inf), r >= 0
}

#endif   // LRC_VARIANTS

// Per caller's triangle ROW: (v0, label bits), (Ng, 0) -- all a known hit needs to give t again.  Read by the multi-GPU
// cloud rebuild only (lrc_cloud_from_prims_dev), so it is built from the slot arrays the first time that is called.
__global__ __launch_bounds__(kBlock) void prim_plane_kernel(const float4* __restrict__ tris, const uint32_t* __restrict__ slot_prim,
                                                            const uint32_t* __restrict__ slot_label, uint32_t num_slots,
                                                            uint32_t num_prims, float4* __restrict__ plane) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= num_slots) return;
    const uint32_t row = slot_prim[k];
    if (row >= num_prims) return;
    const float4 a = tris[(size_t)k * 3], c = tris[(size_t)k * 3 + 2];
    plane[(size_t)row * 2] = make_float4(a.x, a.y, a.z, __uint_as_float(slot_label[k]));
    plane[(size_t)row * 2 + 1] = make_float4(c.y, c.z, c.w, 0.0f);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

// shared with the other translation units of the library (not part of the public ABI)
int lrc_internal_fail(int code, const char* msg) { return fail(code, msg ? msg : ""); }
int lrc_internal_ctx_device(const lrc_ctx* ctx) { return ctx ? ctx->device : 0; }

const char* lrc_version(void) { return "lidarcast 0.1.0 (gfx950)"; }

const char* lrc_last_error(void) { return g_err.c_str(); }

int lrc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int lrc_ctx_create(int device, lrc_ctx** out_ctx) {
    if (!out_ctx) return fail(LRC_ERR_INVALID_ARG, "lrc_ctx_create: out_ctx is NULL");
    *out_ctx = nullptr;
    int n = lrc_device_count();
    if (n <= 0) return fail(LRC_ERR_NO_DEVICE, "lrc_ctx_create: no HIP device is visible");
    if (device < 0 || device >= n)
        return fail(LRC_ERR_NO_DEVICE, "lrc_ctx_create: device ordinal out of range");
    LRC_HIP(hipSetDevice(device));
    lrc_ctx* c = new (std::nothrow) lrc_ctx();
    if (!c) return fail(LRC_ERR_OOM, "lrc_ctx_create: out of host memory");
    c->device = device;
    // the signal word of the launch chain; where stream memory operations are not available launches are simply not chained
    int can_wait = 0;
    if (hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, device) == hipSuccess && can_wait) {
        void* w = nullptr;
        if (hipExtMallocWithFlags(&w, 8, hipMallocSignalMemory) == hipSuccess && w) {
            if (hipStreamWriteValue64(nullptr, w, 0, 0) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess) {
                c->chain_word = (uint64_t*)w;
            } else {
                (void)hipFree(w);
            }
        }
    }
    (void)hipGetLastError();
    *out_ctx = c;
    return LRC_OK;
}

int lrc_ctx_set_launch_chaining(lrc_ctx* ctx, int enabled) {
    if (!ctx) return fail(LRC_ERR_INVALID_ARG, "lrc_ctx_set_launch_chaining: ctx is NULL");
    ctx->chain_enabled = enabled != 0;
    return LRC_OK;
}

int lrc_ctx_get_launch_chaining(const lrc_ctx* ctx, int* enabled, int* supported) {
    if (!ctx) return fail(LRC_ERR_INVALID_ARG, "lrc_ctx_get_launch_chaining: ctx is NULL");
    if (enabled) *enabled = ctx->chain_enabled ? 1 : 0;
    if (supported) *supported = ctx->chain_word ? 1 : 0;
    return LRC_OK;
}

int lrc_ctx_destroy(lrc_ctx* ctx) {
    if (!ctx) return LRC_OK;
    (void)hipSetDevice(ctx->device);
    for (int k = 0; k < kPoolSlots; ++k)
        if (ctx->pool[k]) (void)hipFree(ctx->pool[k]);
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    lrc::arena_release(&ctx->build_arena);
    if (ctx->s_compute) (void)hipStreamDestroy(ctx->s_compute);
    if (ctx->s_copy) (void)hipStreamDestroy(ctx->s_copy);
    if (ctx->s_stats) (void)hipStreamDestroy(ctx->s_stats);
    for (hipEvent_t e : ctx->ev_chunk) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_compact) if (e) (void)hipEventDestroy(e);
    if (ctx->chain_word) (void)hipFree(ctx->chain_word);
    if (ctx->stat_scratch) (void)hipFree(ctx->stat_scratch);
    for (hipEvent_t e : ctx->compact_done) if (e) (void)hipEventDestroy(e);
    for (lrc_ctx::TileScratch* sc : {&ctx->compact_scratch[0], &ctx->compact_scratch[1], &ctx->compact_scratch[2],
                                     &ctx->compact_scratch[3], &ctx->cloud_scratch}) {
        if (sc->d_tile_off) (void)hipFree(sc->d_tile_off);
        if (sc->d_tile_cnt) (void)hipFree(sc->d_tile_cnt);
        if (sc->d_super_total) (void)hipFree(sc->d_super_total);
        if (sc->d_super_base) (void)hipFree(sc->d_super_base);
        if (sc->d_dirs_soa) (void)hipFree(sc->d_dirs_soa);
    }
    delete ctx;
    return LRC_OK;
}

int lrc_ctx_synchronize(lrc_ctx* ctx) {
    if (!ctx) return fail(LRC_ERR_INVALID_ARG, "lrc_ctx_synchronize: ctx is NULL");
    LRC_HIP(hipSetDevice(ctx->device));
    LRC_HIP(hipDeviceSynchronize());
    return LRC_OK;
}

int lrc_scene_destroy(lrc_scene* s) {
    if (!s) return LRC_OK;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    if (s->slab) {
        (void)hipFree(s->slab);
    } else {
        if (s->d_nodes) (void)hipFree(s->d_nodes);
        if (s->d_tris) (void)hipFree(s->d_tris);
        if (s->d_slot_prim) (void)hipFree(s->d_slot_prim);
        if (s->d_slot_label) (void)hipFree(s->d_slot_label);
        if (s->d_slot_box) (void)hipFree(s->d_slot_box);
        if (s->d_nodes_q) (void)hipFree(s->d_nodes_q);
        if (s->d_nodes_n) (void)hipFree(s->d_nodes_n);
    }
    if (s->d_prim_plane) (void)hipFree(s->d_prim_plane);
    if (s->d_slot_sphere) (void)hipFree(s->d_slot_sphere);
    if (s->d_nodes_q4) (void)hipFree(s->d_nodes_q4);
    if (s->d_nodes_n4) (void)hipFree(s->d_nodes_n4);
    delete s;
    return LRC_OK;
}

namespace {

int env_int(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }

void build_options_from_env(lrc::BuildOptions& opt) {
    opt.max_leaf = env_int("LRC_MAX_LEAF", opt.max_leaf);
    opt.bfs_nodes = env_int("LRC_BFS_NODES", opt.bfs_nodes);
    opt.depth_slack = env_int("LRC_DEPTH_SLACK", opt.depth_slack);
    opt.median_only = env_int("LRC_BUILD_MEDIAN_ONLY", 0);      // test hook: every split takes the median fallback
    opt.subtrees = env_int("LRC_BUILD_SUBTREES", opt.subtrees);  // 0: the device builder goes level by level to the bottom (A/B)
}

// LRC_QNODES: 0 = float32 nodes only, 1 (default) = quantised images when the grid is fine enough for the scene's
// triangles, 2 = whenever the grid fits.  The grid has 2^15 cells along each axis of the scene; where the cells are
// not small against the leaf boxes (a very large scene of small triangles) the widened boxes cost more triangle tests
// than the smaller nodes save (measured: DESIGN.md section 4.1).
int qnodes_mode() { return env_int("LRC_QNODES", 1); }      // read at every scene creation

void qnodes_report(const lrc_scene* s, double infl) {
    if (std::getenv("LRC_QNODES_VERBOSE"))
        std::fprintf(stderr, "[qnodes] leaf box inflation %.3f, W = %g %g %g -> %s\n", infl, (double)s->qW[0],
                     (double)s->qW[1], (double)s->qW[2], s->d_nodes_q ? "quantised images" : "float32 nodes");
}

// The scene build on the device (lrc_bvh_device.hip): the same tree, the same bytes as the host builder below, in
// milliseconds.  Returns lrc::kDevBuildUnsupported (1) for a mesh the device path does not take (<= max_leaf triangles).
int scene_create_device(lrc_ctx* ctx, const float* verts3, uint64_t V, const uint32_t* tris3, uint64_t T,
                        const uint16_t* tri_sem, const uint16_t* tri_ins, bool on_device, lrc_scene** out_scene) {
    lrc_scene* s = new (std::nothrow) lrc_scene();
    if (!s) return fail(LRC_ERR_OOM, "lrc_scene_create: out of host memory");
    s->ctx = ctx;
    lrc::DeviceScene d;
    std::string err;
    lrc::BuildOptions opt;
    build_options_from_env(opt);
    const int rc = lrc::build_bvh_device(&ctx->build_arena, verts3, V, tris3, T, tri_sem, tri_ins, on_device, opt,
                                         qnodes_mode(), &d, &err);
    if (rc != lrc::kDevBuildOk) {
        delete s;
        return rc < 0 ? fail(rc, err) : rc;
    }
    s->slab = d.slab;
    s->d_nodes = (float4*)d.nodes;
    s->d_tris = (float4*)d.tris;
    s->d_slot_prim = d.slot_prim;
    s->d_slot_label = d.slot_label;
    s->d_slot_box = d.slot_box;
    s->d_nodes_q = (uint4*)d.nodes_q;
    s->d_nodes_n = (float4*)d.nodes_n;
    for (int a = 0; a < 3; ++a) { s->qbase[a] = d.qbase[a]; s->qW[a] = d.qW[a]; s->qinvW[a] = d.qinvW[a]; }
    lrc_scene_info& in = s->info;
    in.num_vertices = V;
    in.num_triangles = T;
    in.num_nodes = d.num_nodes;
    in.num_leaves = d.num_leaves;
    in.num_slots = d.num_slots;
    in.max_depth = d.max_depth;
    in.max_leaf_size = d.max_leaf_size;
    for (int k = 0; k < 3; ++k) { in.bounds_lo[k] = d.bounds_lo[k]; in.bounds_hi[k] = d.bounds_hi[k]; }
    in.device_bytes = d.slab_bytes;
    in.build_ms = d.ms_hierarchy + d.ms_emit;
    in.upload_ms = d.ms_upload;
    in.quantised_nodes = s->d_nodes_q ? 1u : 0u;
    in.leaf_inflation = (float)d.leaf_inflation;
    in.device_build = 1u;
    qnodes_report(s, d.leaf_inflation);
    *out_scene = s;
    return LRC_OK;
}

int scene_create_host(lrc_ctx* ctx, const float* verts3, uint64_t V, const uint32_t* tris3, uint64_t T,
                      const uint16_t* tri_sem, const uint16_t* tri_ins, lrc_scene** out_scene) {
    for (uint64_t i = 0; i < 3 * T; ++i)
        if (tris3[i] >= V)
            return fail(LRC_ERR_INVALID_ARG, "lrc_scene_create: triangle index out of range");
    for (uint64_t i = 0; i < 3 * V; ++i) {
        uint32_t u;   // bit test: immune to any finite-math assumption of the compiler
        std::memcpy(&u, &verts3[i], 4);
        if ((u & 0x7FFFFFFFu) > 0x49742400u /* 1e6f */)
            return fail(LRC_ERR_INVALID_ARG,
                        "lrc_scene_create: vertex coordinate is not finite or exceeds 1e6");
    }
    lrc_scene* s = new (std::nothrow) lrc_scene();
    if (!s) return fail(LRC_ERR_OOM, "lrc_scene_create: out of host memory");
    s->ctx = ctx;

    lrc::HostBVH h;
    lrc::BuildOptions opt;
    build_options_from_env(opt);
    auto t0 = std::chrono::steady_clock::now();
    try {
        lrc::build_bvh(verts3, V, tris3, T, tri_sem, tri_ins, opt, &h);
    } catch (const std::bad_alloc&) {
        delete s;
        return fail(LRC_ERR_OOM, "lrc_scene_create: out of host memory during the BVH build");
    } catch (...) {
        delete s;
        return fail(LRC_ERR_INTERNAL, "lrc_scene_create: BVH build failed");
    }
    auto t1 = std::chrono::steady_clock::now();

    lrc_scene_info& in = s->info;
    in.num_vertices = V;
    in.num_triangles = T;
    in.num_nodes = h.num_nodes;
    in.num_leaves = h.num_leaves;
    in.num_slots = h.num_slots;
    in.max_depth = h.max_depth;
    in.max_leaf_size = h.max_leaf_size;
    for (int k = 0; k < 3; ++k) { in.bounds_lo[k] = h.bounds_lo[k]; in.bounds_hi[k] = h.bounds_hi[k]; }
    in.build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();

    auto upload = [&](void** dptr, const void* src, size_t bytes) -> int {
        if (!bytes) return LRC_OK;
        LRC_HIP(hipMalloc(dptr, bytes));
        LRC_HIP(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
        in.device_bytes += bytes;
        return LRC_OK;
    };
    int rc;
    if ((rc = upload((void**)&s->d_nodes, h.nodes.data(), h.nodes.size() * 4)) ||
        (rc = upload((void**)&s->d_tris, h.tri_rec.data(), h.tri_rec.size() * 4)) ||
        (rc = upload((void**)&s->d_slot_prim, h.slot_prim.data(), h.slot_prim.size() * 4)) ||
        (rc = upload((void**)&s->d_slot_label, h.slot_label.data(), h.slot_label.size() * 4)) ||
        (rc = upload((void**)&s->d_slot_box, h.slot_box.data(), h.slot_box.size() * 4))) {
        std::string keep = g_err;
        lrc_scene_destroy(s);
        g_err = keep;
        return rc;
    }
    {
        std::vector<uint32_t> q8, q16;
        std::vector<float> n16, n32;
        double infl = 1.0;
        const int mode = qnodes_mode();
        lrc::QGrid g;
        if (mode != 0 && lrc::make_qgrid(h, s->qbase, s->qW, s->qinvW, g) && lrc::build_qnodes(h, g, q8, n16, &infl) &&
            (mode >= 2 || infl <= lrc::kQnodeMaxInflation)) {
#ifdef LRC_VARIANTS
            if (env_int("LRC_WIDE", 0) != 0 && !lrc::build_q4nodes(h, g, q16, n32, &s->num_nodes4)) { q16.clear(); n32.clear(); }
#endif
            if ((rc = upload((void**)&s->d_nodes_q, q8.data(), q8.size() * 4)) ||
                (rc = upload((void**)&s->d_nodes_n, n16.data(), n16.size() * 4)) ||
                (rc = upload((void**)&s->d_nodes_q4, q16.data(), q16.size() * 4)) ||
                (rc = upload((void**)&s->d_nodes_n4, n32.data(), n32.size() * 4))) {
                std::string keep = g_err;
                lrc_scene_destroy(s);
                g_err = keep;
                return rc;
            }
        }
        in.quantised_nodes = s->d_nodes_q ? 1u : 0u;
        in.leaf_inflation = (float)infl;
        qnodes_report(s, infl);
    }
    auto t2 = std::chrono::steady_clock::now();
    in.upload_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    *out_scene = s;
    return LRC_OK;
}

// LRC_DEVICE_BUILD: 1 (default) = build the scene on the GPU, 0 = host builder (same tree, same bytes; the reference
// for the device builder in the tests).  The four-wide images of the laboratory build exist on the host path only.
bool want_device_build() {
#ifdef LRC_VARIANTS
    if (env_int("LRC_WIDE", 0) != 0) return false;
#endif
    return env_int("LRC_DEVICE_BUILD", 1) != 0;
}

}  // namespace

int lrc_scene_create(lrc_ctx* ctx, const float* verts3, uint64_t V, const uint32_t* tris3, uint64_t T,
                     const uint16_t* tri_sem, const uint16_t* tri_ins, lrc_scene** out_scene) {
    if (!out_scene) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_create: out_scene is NULL");
    *out_scene = nullptr;
    if (!ctx) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_create: ctx is NULL");
    if ((V && !verts3) || (T && !tris3))
        return fail(LRC_ERR_INVALID_ARG, "lrc_scene_create: NULL vertex or triangle array");
    if (T >= (1ull << 28) || V >= (1ull << 32))
        return fail(LRC_ERR_INVALID_ARG, "lrc_scene_create: mesh too large (T < 2^28, V < 2^32)");
    LRC_HIP(hipSetDevice(ctx->device));
    if (want_device_build()) {
        const int rc = scene_create_device(ctx, verts3, V, tris3, T, tri_sem, tri_ins, false, out_scene);
        if (rc != lrc::kDevBuildUnsupported) return rc;
    }
    return scene_create_host(ctx, verts3, V, tris3, T, tri_sem, tri_ins, out_scene);
}

int lrc_scene_create_dev(lrc_ctx* ctx, const float* d_verts3, uint64_t V, const uint32_t* d_tris3, uint64_t T,
                         const uint16_t* d_tri_sem, const uint16_t* d_tri_ins, lrc_scene** out_scene) {
    if (!out_scene) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_create_dev: out_scene is NULL");
    *out_scene = nullptr;
    if (!ctx) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_create_dev: ctx is NULL");
    if ((V && !d_verts3) || (T && !d_tris3))
        return fail(LRC_ERR_INVALID_ARG, "lrc_scene_create_dev: NULL vertex or triangle array");
    if (T >= (1ull << 28) || V >= (1ull << 32))
        return fail(LRC_ERR_INVALID_ARG, "lrc_scene_create_dev: mesh too large (T < 2^28, V < 2^32)");
    LRC_HIP(hipSetDevice(ctx->device));
    if (want_device_build()) {
        const int rc = scene_create_device(ctx, d_verts3, V, d_tris3, T, d_tri_sem, d_tri_ins, true, out_scene);
        if (rc != lrc::kDevBuildUnsupported) return rc;
    }
    // a mesh the device builder does not take (a handful of triangles), or LRC_DEVICE_BUILD=0: build on the host
    std::vector<float> hv(3 * V);
    std::vector<uint32_t> ht(3 * T);
    std::vector<uint16_t> hs(d_tri_sem ? T : 0), hi(d_tri_ins ? T : 0);
    if (V) LRC_HIP(hipMemcpy(hv.data(), d_verts3, 3 * V * 4, hipMemcpyDeviceToHost));
    if (T) LRC_HIP(hipMemcpy(ht.data(), d_tris3, 3 * T * 4, hipMemcpyDeviceToHost));
    if (d_tri_sem && T) LRC_HIP(hipMemcpy(hs.data(), d_tri_sem, T * 2, hipMemcpyDeviceToHost));
    if (d_tri_ins && T) LRC_HIP(hipMemcpy(hi.data(), d_tri_ins, T * 2, hipMemcpyDeviceToHost));
    return scene_create_host(ctx, hv.data(), V, ht.data(), T, d_tri_sem ? hs.data() : nullptr,
                             d_tri_ins ? hi.data() : nullptr, out_scene);
}

int lrc_scene_get_info(const lrc_scene* scene, lrc_scene_info* out_info) {
    if (!scene || !out_info) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_get_info: NULL argument");
    *out_info = scene->info;
    return LRC_OK;
}

int lrc_scene_set_options(lrc_scene* scene, const lrc_scan_options* opts) {
    if (!scene) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_set_options: scene is NULL");
    if (!opts) { scene->opts = lrc_scan_options{}; return LRC_OK; }
    if (!(opts->min_range >= 0.0) || (opts->incident_mode != 0 && opts->incident_mode != 1))
        return fail(LRC_ERR_INVALID_ARG, "lrc_scene_set_options: min_range < 0 or unknown incident_mode");
    scene->opts = *opts;
    return LRC_OK;
}

int lrc_scene_get_counters(const lrc_scene* scene, uint64_t* launches, uint64_t* rays) {
    if (!scene) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_get_counters: scene is NULL");
    if (launches) *launches = scene->launches;
    if (rays) *rays = scene->rays;
    return LRC_OK;
}

int lrc_scene_export_bvh(const lrc_scene* s, float* nodes16, uint32_t* slot_prim) {
    if (!s) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_export_bvh: scene is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    if (nodes16 && s->info.num_nodes)
        LRC_HIP(hipMemcpy(nodes16, s->d_nodes, s->info.num_nodes * 64, hipMemcpyDeviceToHost));
    if (slot_prim && s->info.num_slots)
        LRC_HIP(hipMemcpy(slot_prim, s->d_slot_prim, s->info.num_slots * 4, hipMemcpyDeviceToHost));
    return LRC_OK;
}

// the plane table of the cloud rebuild, built on first use.  The table is complete before the pointer is published (one
// synchronisation, once per scene): a second caller -- another host thread, or another non-blocking stream -- either sees
// NULL and waits for the mutex, or sees a finished table; it can never read a zeroed or half-written one.
static int ensure_prim_plane(lrc_scene* s, hipStream_t st) {
    if (s->info.num_triangles == 0) return LRC_OK;
    if (__atomic_load_n(&s->d_prim_plane, __ATOMIC_ACQUIRE)) return LRC_OK;
    std::lock_guard<std::mutex> lock(s->plane_mutex);
    if (s->d_prim_plane) return LRC_OK;
    const uint64_t T = s->info.num_triangles;
    float4* table = nullptr;
    LRC_HIP(hipMalloc((void**)&table, T * 32));
    hipError_t e = hipMemsetAsync(table, 0, T * 32, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(prim_plane_kernel, dim3((uint32_t)((s->info.num_slots + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           (const float4*)s->d_tris, (const uint32_t*)s->d_slot_prim, (const uint32_t*)s->d_slot_label,
                           (uint32_t)s->info.num_slots, (uint32_t)T, table);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipFree(table);
        (void)hipGetLastError();
        return fail(LRC_ERR_HIP, std::string("plane table of the cloud rebuild: ") + hipGetErrorString(e));
    }
    s->info.device_bytes += T * 32;
    __atomic_store_n(&s->d_prim_plane, table, __ATOMIC_RELEASE);
    return LRC_OK;
}

int lrc_scene_export_array(const lrc_scene* cs, int which, void* dst, uint64_t dst_bytes, uint64_t* out_bytes) {
    if (!cs) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_export_array: scene is NULL");
    lrc_scene* s = const_cast<lrc_scene*>(cs);
    if (which == LRC_ARRAY_PRIM_PLANE) {
        LRC_HIP(hipSetDevice(s->ctx->device));
        int rc = ensure_prim_plane(s, nullptr);
        if (rc) return rc;
        LRC_HIP(hipStreamSynchronize(nullptr));
    }
    const void* src = nullptr;
    uint64_t bytes = 0;
    const lrc_scene_info& in = s->info;
    switch (which) {
        case LRC_ARRAY_NODES: src = s->d_nodes; bytes = in.num_nodes * 64; break;
        case LRC_ARRAY_TRIS: src = s->d_tris; bytes = in.num_slots * 48; break;
        case LRC_ARRAY_SLOT_PRIM: src = s->d_slot_prim; bytes = in.num_slots * 4; break;
        case LRC_ARRAY_SLOT_LABEL: src = s->d_slot_label; bytes = in.num_slots * 4; break;
        case LRC_ARRAY_PRIM_PLANE: src = s->d_prim_plane; bytes = in.num_triangles * 32; break;
        case LRC_ARRAY_NODES_Q: src = s->d_nodes_q; bytes = s->d_nodes_q ? in.num_nodes * 32 : 0; break;
        case LRC_ARRAY_NODES_N: src = s->d_nodes_n; bytes = s->d_nodes_n ? in.num_nodes * 64 : 0; break;
        default: return fail(LRC_ERR_INVALID_ARG, "lrc_scene_export_array: unknown array");
    }
    if (out_bytes) *out_bytes = bytes;
    if (!dst || !bytes) return LRC_OK;
    if (dst_bytes < bytes) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_export_array: destination too small");
    LRC_HIP(hipSetDevice(s->ctx->device));
    LRC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return LRC_OK;
}

#ifdef LRC_VARIANTS
// Laboratory build (-DLRC_VARIANTS; tests/test_parity_gpu.py::test_kernel_variants_are_bit_identical, tools/): the measured
// alternatives of DESIGN.md section 4.1, selected by environment variables.  Every one returns the product kernel's bytes.
// Returns 1 when no variant is selected (the caller goes on with the product dispatch).
static int launch_trace_lab(lrc_scene* s, TraceParams& p, int gen, hipStream_t st, bool stats, uint64_t nblk, size_t lds,
                            uint32_t depth) {
    static const int force_redo = env_int("LRC_DEBUG_FORCE_REDO", 0);
    static const int leafw = env_int("LRC_LEAFW", kLeafW) == 1 ? 1 : 2, uni = env_int("LRC_UNIFORM", 1), spec = env_int("LRC_SPEC", 0);
    static const int sector = env_int("LRC_SECTOR", 1), refill = env_int("LRC_REFILL", 0);
    p.force_redo = (uint32_t)force_redo;
    const bool qn = s->d_nodes_q != nullptr, wide = s->d_nodes_q4 != nullptr;
#define LRC_LAB(G, W, U, S, Q) \
    hipLaunchKernelGGL((trace_kernel<G, W, U, S, false, Q>), dim3((uint32_t)nblk), dim3(kTBlock), lds, st, p)
#define LRC_LAB_PICK(G)                                                                              \
    do {                                                                                             \
        if (spec) { if (leafw == 2) LRC_LAB(G, 2, true, true, 0); else LRC_LAB(G, 1, true, true, 0); } \
        else if (!uni) { if (leafw == 2) LRC_LAB(G, 2, false, false, 0); else LRC_LAB(G, 1, false, false, 0); } \
        else { if (qn) LRC_LAB(G, 1, true, false, 1); else LRC_LAB(G, 1, true, false, 0); }     /* leafw == 1 */ \
    } while (0)
    if (gen == 3 && sector && !stats) {
        const lrc_grid& g = *s->cur_grid;
        SectorParams q{};
        { int rc = ensure_prim_plane(s, st); if (rc) return rc; }      // the packet kernel keys rays by triangle row
        p.prim_plane = s->d_prim_plane;
        q.tp = p;
        if (!s->d_slot_sphere && s->info.num_slots) {
            LRC_HIP(hipMalloc((void**)&s->d_slot_sphere, s->info.num_slots * 16));
            s->info.device_bytes += s->info.num_slots * 16;
            hipLaunchKernelGGL(slot_sphere_kernel, dim3((uint32_t)((s->info.num_slots + kBlock - 1) / kBlock)), dim3(kBlock),
                               0, st, (const float*)s->d_slot_box, (uint32_t)s->info.num_slots, s->d_slot_sphere);
        }
        q.slot_sphere = s->d_slot_sphere;
        q.H = g.lines; q.W = g.width;
        static const int nl_env = env_int("LRC_SECTOR_LINES", 0);
        uint32_t nl = nl_env > 0 ? (uint32_t)nl_env : 8u;
        if (nl > 8u) nl = 8u;
        if (nl > g.lines) nl = g.lines;
        q.nl = nl;
        q.groups = (g.lines + nl - 1) / nl;
        q.az0 = (float)g.az0; q.az_step = (float)g.az_step;
        q.levels = s->info.max_depth + 1u;
        q.stack_cap = 128u * q.levels;
        const uint64_t P = p.total / p.rays_per_pose;
        q.num_packets = P * q.groups * (g.width / 64u);
        if (q.num_packets > 0x7FFFFFFFull) return fail(LRC_ERR_INVALID_ARG, "too many rays for one launch");
        const size_t R = (size_t)nl * 64;
        const size_t lds_s = R * 8 + R * 12 + (size_t)q.levels * 4 + (size_t)q.stack_cap * 4 + kLeafQ * 4 + kPairQ * 8;
        static const int sdiag = env_int("LRC_SECTOR_DIAG", 0);
        if (sdiag) {      // work totals of this launch, printed to stderr (tools only)
            unsigned long long* d = nullptr;
            LRC_HIP(hipMalloc((void**)&d, kSectorDiagWords * 8));
            LRC_HIP(hipMemsetAsync(d, 0, kSectorDiagWords * 8, st));
            q.diag = d;
            hipLaunchKernelGGL(sector_kernel<true>, dim3((uint32_t)q.num_packets), dim3(64), lds_s, st, q);
            unsigned long long h[kSectorDiagWords];
            LRC_HIP(hipStreamSynchronize(st));
            LRC_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
            (void)hipFree(d);
            std::fprintf(stderr, "[sector diag] packets %llu lines/packet %u: node rounds %llu, nodes %llu, leaves %llu, "
                                 "triangles %llu, pairs %llu, pair rounds %llu, candidate rays %llu, accepted %llu\n",
                         (unsigned long long)q.num_packets, nl, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
        } else {
            hipLaunchKernelGGL(sector_kernel<false>, dim3((uint32_t)q.num_packets), dim3(64), lds_s, st, q);
        }
        return LRC_OK;
    }
    if (gen == 3) gen = 1;
    if (refill > 1 && gen == 1 && !stats && p.rays_per_pose % (64u * (refill >= 4 ? 4 : 2)) == 0) {
        const int K = refill >= 4 ? 4 : 2;        // K rays per lane with private refill
        const uint64_t nb = (p.total + 64ull * K - 1) / (64ull * K);
        const size_t ldsK = ((size_t)depth * 64 + (size_t)K * 64 * 2) * sizeof(int);
        static const int rw = env_int("LRC_REFILL_W", 0);
#define LRC_RF(KK, WW) hipLaunchKernelGGL((trace_refill_kernel<KK, WW>), dim3((uint32_t)nb), dim3(kTBlock), ldsK, st, p, depth)
        if (K == 4) { if (rw >= 7) LRC_RF(4, 7); else LRC_RF(4, 5); }
        else { if (rw >= 7) LRC_RF(2, 7); else LRC_RF(2, 6); }
#undef LRC_RF
        return LRC_OK;
    }
    if (stats) {
        if (gen == 1 && qn && wide) { hipLaunchKernelGGL((trace_kernel<1, kLeafW, true, false, true, 2>), dim3((uint32_t)nblk), dim3(kTBlock), lds, st, p); return LRC_OK; }
        return 1;
    }
    const bool plain = leafw == kLeafW && uni && !spec && !(qn && wide);
    if (plain || gen == 2) return 1;
    const bool wide_only = qn && wide && leafw == kLeafW && uni && !spec;
    if (gen == 1) { if (wide_only) LRC_LAB(1, kLeafW, true, false, 2); else LRC_LAB_PICK(1); }
    else { if (wide_only) LRC_LAB(0, kLeafW, true, false, 2); else LRC_LAB_PICK(0); }
#undef LRC_LAB_PICK
#undef LRC_LAB
    return LRC_OK;
}
#endif   // LRC_VARIANTS

// One launch of the trace kernel over p.total rays.  gen: 0 explicit rays, 1 pose x direction table, 2 pose x scan angles,
// 3 a grid scan (lrc_scan_grid_*: the per-ray kernel here; the packet kernel in the laboratory build).
static int launch_trace(lrc_scene* s, TraceParams& p, int gen, hipStream_t st, bool stats = false) {
    p.nodes = s->d_nodes;
    p.tris = s->d_tris;
    p.slot_prim = s->d_slot_prim;
    p.slot_label = s->d_slot_label;
    p.slot_box = s->d_slot_box;
    p.prim_plane = s->d_prim_plane;
    p.num_nodes = (uint32_t)s->info.num_nodes;
    p.nodes_q = s->d_nodes_q;
    p.nodes_n = s->d_nodes_n;
    p.nodes_q4 = s->d_nodes_q4;
    p.nodes_n4 = s->d_nodes_n4;
    for (int a = 0; a < 3; ++a) { p.qbase[a] = s->qbase[a]; p.qW[a] = s->qW[a]; p.qinvW[a] = s->qinvW[a]; }
    const bool qn = s->d_nodes_q != nullptr;
    p.min_range = s->opts.min_range;
    p.incident_mode = s->opts.incident_mode;
    if (!p.range_noise && s->opts.range_noise) {      // device entry points: the pointer is a device pointer
        if (s->opts.range_noise_len != p.total)
            return fail(LRC_ERR_INVALID_ARG, "range_noise_len does not match the number of rays of this call");
        p.range_noise = s->opts.range_noise;
    }
    if (p.total == 0) return LRC_OK;
    const uint64_t nblk = (p.total + kTBlock - 1) / kTBlock;
    if (nblk + p.pre.blocks > 0x7FFFFFFFull) return fail(LRC_ERR_INVALID_ARG, "too many rays for one launch");
    // stack entries needed = deepest leaf depth (one pending sibling per inner level above it)
    const uint32_t depth = s->info.max_depth < 1 ? 1 : s->info.max_depth;
    const size_t lds = (size_t)depth * kTBlock * sizeof(int);
    p.stack_cap = depth;
    // Workgroup -> tile order.  A pose-batched scan deals every pose's tiles to the 8 XCDs in 16 chunks: each XCD works on
    // every pose (the poses of a trajectory cost differently: balance) but always on the same sixteenth-pairs of the scan
    // pattern, so what its L2 holds after one pose is what the next pose needs.  Other launches, and
    // scans whose pose is not 16 chunks of a power-of-two >= 16 tiles, keep the eight contiguous tile ranges.
    p.tile_chunk_log2 = 0;
    // ... while the scene is resident in the 256 MiB Infinity Cache.  Past it (synth_hall: 0.3 GB) an L2 miss costs an HBM
    // round trip and the XCD-local pose run -- neighbouring scan lines of ONE pose share their subtrees in the XCD's L2 --
    // is worth more than the balance: measured +1...+16 % for the striped order there, -1...-10 % on the cache-resident
    // scenes (profiles/r03_xcd_striping_sweep.txt).
    constexpr uint64_t kStripeSceneBytes = 192ull << 20;
    if (gen == 1 && p.rays_per_pose % 64 == 0 && s->info.device_bytes <= kStripeSceneBytes) {
        const uint64_t tpp = p.rays_per_pose / 64;
        if (tpp % 16 == 0) {
            const uint64_t chunk = tpp / 16;
            if (chunk >= 16 && (chunk & (chunk - 1)) == 0) { uint32_t k = 0; while ((1ull << k) < chunk) ++k; p.tile_chunk_log2 = k + 1; }
        }
    }
#ifdef LRC_VARIANTS
    {   // A/B knob (tools/chunk_sweep*.sh): LRC_TILE_CHUNK = n (power of two) tiles per chunk, -1 = contiguous ranges
        static const int tc = env_int("LRC_TILE_CHUNK", 0);
        if (tc < 0) p.tile_chunk_log2 = 0;
        if (tc > 0 && (tc & (tc - 1)) == 0) { uint32_t k = 0; while ((1 << k) < tc) ++k; p.tile_chunk_log2 = k + 1; }
    }
#endif
#ifdef LRC_VARIANTS
    {
        const int rc = launch_trace_lab(s, p, gen, st, stats, nblk, lds, depth);
        if (rc <= 0) {
            if (rc == LRC_OK) { LRC_HIP(hipGetLastError()); s->launches += 1; s->rays += p.total; }
            return rc;
        }
    }
#endif
    if (gen == 3) gen = 1;
    // Launch chain.  Every product trace launch signs the context's signal word when its last workgroup starts.  A launch
    // that goes to ANOTHER stream than the previous one is held until that signature is there: the caller keeps two scans
    // in flight (two streams, two record sets), so instead of both launches dispatching side by side and ending together --
    // or, by chance, not -- this one starts exactly when the previous one has no workgroup left to hand out, and its waves
    // fill the slots the previous launch's long-running last waves leave empty.  Same stream as before: stream order already
    // serialises the two launches, nothing is added.  The wait only ever refers to a launch enqueued earlier, so it cannot
    // close a cycle with the caller's own events.
    lrc_ctx* const cx_ = s->ctx;
    if (cx_->chain_word && cx_->chain_enabled) {
        if (cx_->chain_seq > 0 && st != cx_->chain_stream && st != nullptr && cx_->chain_stream != nullptr)
            LRC_HIP(hipStreamWaitValue64(st, cx_->chain_word, cx_->chain_seq, hipStreamWaitValueGte, ~0ull));
        p.chain_word = cx_->chain_word;
        p.chain_seq = cx_->chain_seq + 1;
    } else {
        p.chain_word = nullptr;
        p.chain_seq = 0;
    }
#define LRC_LAUNCH(G, S, Q) \
    hipLaunchKernelGGL((trace_kernel<G, kLeafW, true, false, S, Q>), dim3((uint32_t)(nblk + p.pre.blocks)), dim3(kTBlock), lds, st, p)
    if (stats) {   // per-ray traversal counters (lrc_debug_scan_stats)
        if (gen == 1) { if (qn) LRC_LAUNCH(1, true, 1); else LRC_LAUNCH(1, true, 0); }
        else if (gen == 0) { if (qn) LRC_LAUNCH(0, true, 1); else LRC_LAUNCH(0, true, 0); }
        else return fail(LRC_ERR_INVALID_ARG, "traversal statistics are not available for the scan-angle generator");
    } else if (gen == 1) { if (qn) LRC_LAUNCH(1, false, 1); else LRC_LAUNCH(1, false, 0); }
    else if (gen == 2) { if (qn) LRC_LAUNCH(2, false, 1); else LRC_LAUNCH(2, false, 0); }
    else { if (qn) LRC_LAUNCH(0, false, 1); else LRC_LAUNCH(0, false, 0); }
#undef LRC_LAUNCH
    LRC_HIP(hipGetLastError());
    if (p.chain_word) { cx_->chain_seq = p.chain_seq; cx_->chain_stream = st; }
    s->launches += 1;
    s->rays += p.total;
    return LRC_OK;
}

int lrc_scene_get_occupancy(const lrc_scene* s, int* waves_per_cu, int* vgprs, int* lds_bytes) {
    if (!s) return fail(LRC_ERR_INVALID_ARG, "lrc_scene_get_occupancy: scene is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    const uint32_t depth = s->info.max_depth < 1 ? 1 : s->info.max_depth;
    const size_t lds = (size_t)depth * kTBlock * sizeof(int);
    int blocks = 0;
    hipFuncAttributes attr;
    if (s->d_nodes_q) {
        LRC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trace_kernel<1, kLeafW, true, false, false, 1>, kTBlock, lds));
        LRC_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&trace_kernel<1, kLeafW, true, false, false, 1>)));
    } else {
        LRC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trace_kernel<1, kLeafW, true, false, false>, kTBlock, lds));
        LRC_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&trace_kernel<1, kLeafW, true, false, false>)));
    }
    if (waves_per_cu) *waves_per_cu = blocks * (kTBlock / 64);
    if (vgprs) *vgprs = attr.numRegs;
    if (lds_bytes) *lds_bytes = (int)lds;
    return LRC_OK;
}

int lrc_cast_dev(lrc_scene* s, const float* d_rays6, uint64_t n, const double* center3,
                 double max_range, const lrc_hits* d_out, void* stream) {
    if (!s || !d_out) return fail(LRC_ERR_INVALID_ARG, "lrc_cast_dev: NULL scene or output");
    if (n && !d_rays6) return fail(LRC_ERR_INVALID_ARG, "lrc_cast_dev: rays6 is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    TraceParams p{};
    p.rays6 = d_rays6;
    p.total = n;
    p.rays_per_pose = n ? n : 1;
    p.has_center = center3 != nullptr;
    if (center3) { p.cx = center3[0]; p.cy = center3[1]; p.cz = center3[2]; }
    p.max_range = max_range;
    p.out = *d_out;
    return launch_trace(s, p, 0, (hipStream_t)stream);
}

int lrc_cast_segments_dev(lrc_scene* s, const float* d_rays6, uint64_t n, const uint64_t* d_seg_offsets,
                          uint64_t num_segments, const double* d_centers3, double max_range,
                          const lrc_hits* d_out, void* stream) {
    if (!s || !d_out) return fail(LRC_ERR_INVALID_ARG, "lrc_cast_segments_dev: NULL scene or output");
    if (n && (!d_rays6 || !d_seg_offsets || !d_centers3 || num_segments == 0))
        return fail(LRC_ERR_INVALID_ARG, "lrc_cast_segments_dev: NULL rays, offsets or centres");
    if (num_segments > 0x7FFFFFFFull) return fail(LRC_ERR_INVALID_ARG, "lrc_cast_segments_dev: too many segments");
    LRC_HIP(hipSetDevice(s->ctx->device));
    TraceParams p{};
    p.rays6 = d_rays6;
    p.seg_offsets = d_seg_offsets;
    p.seg_centers3 = d_centers3;
    p.num_segments = (uint32_t)num_segments;
    p.total = n;
    p.rays_per_pose = n ? n : 1;
    p.has_center = 1;
    p.max_range = max_range;
    p.out = *d_out;
    return launch_trace(s, p, 0, (hipStream_t)stream);
}

int lrc_scan_poses_dev(lrc_scene* s, const double* d_poses16, uint64_t P, const double* d_dirs3,
                       uint64_t N, double max_range, const lrc_hits* d_out, void* stream) {
    if (!s || !d_out) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_poses_dev: NULL scene or output");
    if (P && N && (!d_poses16 || !d_dirs3))
        return fail(LRC_ERR_INVALID_ARG, "lrc_scan_poses_dev: poses16 or dirs3 is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    TraceParams p{};
    p.poses16 = d_poses16;
    p.dirs3 = d_dirs3;
    p.rays_per_pose = N ? N : 1;
    p.total = P * N;
    p.has_center = 1;
    p.max_range = max_range;
    p.out = *d_out;
    return launch_trace(s, p, 1, (hipStream_t)stream);
}

static int check_grid(const char* who, const lrc_grid* g, uint64_t N) {
    auto bad = [&](const char* m) { return fail(LRC_ERR_INVALID_ARG, std::string(who) + ": " + m); };
    if (!g) return bad("grid is NULL");
    if (g->lines == 0 || g->width == 0 || (uint64_t)g->lines * g->width != N) return bad("lines * width != rays_per_pose");
    if (g->width % 64 || g->width < 256) return bad("the packet kernel needs width % 64 == 0 and width >= 256");
    if (!(std::fabs(g->az_step) > 0.0) || std::fabs(std::fabs(g->az_step) * g->width - 6.283185307179586) > 1e-6)
        return bad("az_step must cover one turn: |az_step| * width = 2 pi");
    return LRC_OK;
}

int lrc_scan_grid_dev(lrc_scene* s, const double* d_poses16, uint64_t P, const double* d_dirs3, const lrc_grid* grid,
                      double max_range, const lrc_hits* d_out, void* stream) {
    if (!s || !d_out) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_grid_dev: NULL scene or output");
    if (!grid) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_grid_dev: grid is NULL");
    const uint64_t N = (uint64_t)grid->lines * grid->width;
    int rc = check_grid("lrc_scan_grid_dev", grid, N);
    if (rc) return rc;
    if (P && (!d_poses16 || !d_dirs3)) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_grid_dev: poses16 or dirs3 is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    TraceParams p{};
    p.poses16 = d_poses16;
    p.dirs3 = d_dirs3;
    p.rays_per_pose = N;
    p.total = P * N;
    p.has_center = 1;
    p.max_range = max_range;
    p.out = *d_out;
    s->cur_grid = grid;
    rc = launch_trace(s, p, 3, (hipStream_t)stream);
    s->cur_grid = nullptr;
    return rc;
}

// ---- host-pointer convenience wrappers ---------------------------------------------------------
namespace {
struct DevBuf {
    void* p = nullptr;
    bool pooled = false;
    ~DevBuf() { if (p && !pooled) (void)hipFree(p); }
    // buffer of the context's staging pool (kept for the next call)
    int get(lrc_ctx* ctx, int slot, size_t bytes) {
        if (ctx->pool_cap[slot] < bytes) {
            if (ctx->pool[slot]) { (void)hipFree(ctx->pool[slot]); ctx->pool[slot] = nullptr; ctx->pool_cap[slot] = 0; }
            const size_t cap = bytes + bytes / 8;
            LRC_HIP(hipMalloc(&ctx->pool[slot], cap));
            ctx->pool_cap[slot] = cap;
        }
        p = ctx->pool[slot];
        pooled = true;
        return LRC_OK;
    }
};
struct HitsStage {
    DevBuf t, prim, normal3, point3, sem, ins, inc, inten;
    lrc_hits d{};
    int alloc(lrc_ctx* ctx, const lrc_hits& h, uint64_t n) {
        if (!n) return LRC_OK;
        int rc;
        if (h.t) { if ((rc = t.get(ctx, kPoolT, n * 4))) return rc; d.t = (float*)t.p; }
        if (h.prim) { if ((rc = prim.get(ctx, kPoolPrim, n * 4))) return rc; d.prim = (uint32_t*)prim.p; }
        if (h.normal3) { if ((rc = normal3.get(ctx, kPoolNormal, n * 12))) return rc; d.normal3 = (float*)normal3.p; }
        if (h.point3) { if ((rc = point3.get(ctx, kPoolPoint, n * 12))) return rc; d.point3 = (float*)point3.p; }
        if (h.sem) { if ((rc = sem.get(ctx, kPoolSem, n * 2))) return rc; d.sem = (uint16_t*)sem.p; }
        if (h.ins) { if ((rc = ins.get(ctx, kPoolIns, n * 2))) return rc; d.ins = (uint16_t*)ins.p; }
        if (h.incident_deg) { if ((rc = inc.get(ctx, kPoolInc, n * 8))) return rc; d.incident_deg = (double*)inc.p; }
        if (h.intensity) { if ((rc = inten.get(ctx, kPoolInten, n * 4))) return rc; d.intensity = (float*)inten.p; }
        return LRC_OK;
    }
    int download(const lrc_hits& h, uint64_t n) {
        if (!n) return LRC_OK;
        if (h.t) LRC_HIP(hipMemcpy(h.t, d.t, n * 4, hipMemcpyDeviceToHost));
        if (h.prim) LRC_HIP(hipMemcpy(h.prim, d.prim, n * 4, hipMemcpyDeviceToHost));
        if (h.normal3) LRC_HIP(hipMemcpy(h.normal3, d.normal3, n * 12, hipMemcpyDeviceToHost));
        if (h.point3) LRC_HIP(hipMemcpy(h.point3, d.point3, n * 12, hipMemcpyDeviceToHost));
        if (h.sem) LRC_HIP(hipMemcpy(h.sem, d.sem, n * 2, hipMemcpyDeviceToHost));
        if (h.ins) LRC_HIP(hipMemcpy(h.ins, d.ins, n * 2, hipMemcpyDeviceToHost));
        if (h.incident_deg) LRC_HIP(hipMemcpy(h.incident_deg, d.incident_deg, n * 8, hipMemcpyDeviceToHost));
        if (h.intensity) LRC_HIP(hipMemcpy(h.intensity, d.intensity, n * 4, hipMemcpyDeviceToHost));
        return LRC_OK;
    }
};
// host entry points: lrc_scan_options.range_noise is a HOST array; stage it in HBM for the call
struct NoiseStage {
    lrc_scene* s = nullptr;
    const float* host = nullptr;
    DevBuf buf;
    int begin(lrc_scene* scene, uint64_t n) {
        if (!scene->opts.range_noise) return LRC_OK;
        if (scene->opts.range_noise_len != n)
            return fail(LRC_ERR_INVALID_ARG, "range_noise_len does not match the number of rays of this call");
        int rc = buf.get(scene->ctx, kPoolNoise, n * 4);
        if (rc) return rc;
        LRC_HIP(hipMemcpy(buf.p, scene->opts.range_noise, n * 4, hipMemcpyHostToDevice));
        s = scene;
        host = scene->opts.range_noise;
        scene->opts.range_noise = (const float*)buf.p;
        return LRC_OK;
    }
    ~NoiseStage() { if (s) s->opts.range_noise = host; }
};
}  // namespace

int lrc_cast(lrc_scene* s, const float* rays6, uint64_t n, const double* center3, double max_range,
             const lrc_hits* out) {
    if (!s || !out) return fail(LRC_ERR_INVALID_ARG, "lrc_cast: NULL scene or output");
    if (n && !rays6) return fail(LRC_ERR_INVALID_ARG, "lrc_cast: rays6 is NULL");
    if (!n) return LRC_OK;
    LRC_HIP(hipSetDevice(s->ctx->device));
    DevBuf rays;
    int rc = rays.get(s->ctx, kPoolRays, n * 24);
    if (rc) return rc;
    LRC_HIP(hipMemcpy(rays.p, rays6, n * 24, hipMemcpyHostToDevice));
    HitsStage st;
    if ((rc = st.alloc(s->ctx, *out, n))) return rc;
    NoiseStage ns;
    if ((rc = ns.begin(s, n))) return rc;
    rc = lrc_cast_dev(s, (const float*)rays.p, n, center3, max_range, &st.d, nullptr);
    if (rc) return rc;
    LRC_HIP(hipDeviceSynchronize());
    return st.download(*out, n);
}

int lrc_cast_segments(lrc_scene* s, const float* rays6, uint64_t n, const uint64_t* seg_offsets,
                      uint64_t num_segments, const double* centers3, double max_range, const lrc_hits* out) {
    if (!s || !out) return fail(LRC_ERR_INVALID_ARG, "lrc_cast_segments: NULL scene or output");
    if (!n) return LRC_OK;
    if (!rays6 || !seg_offsets || !centers3 || num_segments == 0)
        return fail(LRC_ERR_INVALID_ARG, "lrc_cast_segments: NULL rays, offsets or centres");
    if (seg_offsets[0] != 0 || seg_offsets[num_segments] != n)
        return fail(LRC_ERR_INVALID_ARG, "lrc_cast_segments: offsets must start at 0 and end at num_rays");
    for (uint64_t k = 0; k < num_segments; ++k)
        if (seg_offsets[k] > seg_offsets[k + 1])
            return fail(LRC_ERR_INVALID_ARG, "lrc_cast_segments: offsets must be non-decreasing");
    LRC_HIP(hipSetDevice(s->ctx->device));
    DevBuf rays, offs, cen;
    int rc;
    if ((rc = rays.get(s->ctx, kPoolRays, n * 24)) || (rc = offs.get(s->ctx, kPoolOffs, (num_segments + 1) * 8)) ||
        (rc = cen.get(s->ctx, kPoolCen, num_segments * 24)))
        return rc;
    LRC_HIP(hipMemcpy(rays.p, rays6, n * 24, hipMemcpyHostToDevice));
    LRC_HIP(hipMemcpy(offs.p, seg_offsets, (num_segments + 1) * 8, hipMemcpyHostToDevice));
    LRC_HIP(hipMemcpy(cen.p, centers3, num_segments * 24, hipMemcpyHostToDevice));
    HitsStage st;
    if ((rc = st.alloc(s->ctx, *out, n))) return rc;
    NoiseStage ns;
    if ((rc = ns.begin(s, n))) return rc;
    rc = lrc_cast_segments_dev(s, (const float*)rays.p, n, (const uint64_t*)offs.p, num_segments,
                               (const double*)cen.p, max_range, &st.d, nullptr);
    if (rc) return rc;
    LRC_HIP(hipDeviceSynchronize());
    return st.download(*out, n);
}

int lrc_scan_poses(lrc_scene* s, const double* poses16, uint64_t P, const double* dirs3, uint64_t N,
                   double max_range, const lrc_hits* out) {
    if (!s || !out) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_poses: NULL scene or output");
    if (P && N && (!poses16 || !dirs3))
        return fail(LRC_ERR_INVALID_ARG, "lrc_scan_poses: poses16 or dirs3 is NULL");
    const uint64_t n = P * N;
    if (!n) return LRC_OK;
    LRC_HIP(hipSetDevice(s->ctx->device));
    DevBuf dp, dd;
    int rc;
    if ((rc = dp.get(s->ctx, kPoolPoses, P * 128)) || (rc = dd.get(s->ctx, kPoolDirs, N * 24))) return rc;
    LRC_HIP(hipMemcpy(dp.p, poses16, P * 128, hipMemcpyHostToDevice));
    LRC_HIP(hipMemcpy(dd.p, dirs3, N * 24, hipMemcpyHostToDevice));
    HitsStage st;
    if ((rc = st.alloc(s->ctx, *out, n))) return rc;
    NoiseStage ns;
    if ((rc = ns.begin(s, n))) return rc;
    rc = lrc_scan_poses_dev(s, (const double*)dp.p, P, (const double*)dd.p, N, max_range, &st.d, nullptr);
    if (rc) return rc;
    LRC_HIP(hipDeviceSynchronize());
    return st.download(*out, n);
}

static int ensure_tile_scratch(lrc_ctx* ctx, lrc_ctx::TileScratch& sc, uint64_t ntiles);
static int compact_scratch_for(lrc_ctx* ctx, hipStream_t st, uint64_t ntiles, int* out_set);

// the launches of a compaction on stream `st` with scratch set `sc` (sized by the caller for nseg * ceil(seg_len / 64) tiles)
static int enqueue_compaction(lrc_ctx::TileScratch& sc, uint64_t nseg, uint64_t seg_len, const lrc_compact_io* io, hipStream_t st) {
    const uint64_t tps = (seg_len + 63) / 64;
    const uint64_t ntiles = nseg * tps;
    const uint64_t nblocks = (ntiles + kBlock / 64 - 1) / (kBlock / 64);
    // the trace kernel can hand over its per-wave keep counts (lrc_hits.tile_count) when tiles line up
    const uint32_t* cnt = (io->tile_count && seg_len % 64 == 0) ? io->tile_count : nullptr;
    if (!cnt) {
        hipLaunchKernelGGL(compact_count_kernel, dim3((uint32_t)nblocks), dim3(kBlock), 0, st, io->t, seg_len,
                           tps, ntiles, sc.d_tile_cnt);
        cnt = sc.d_tile_cnt;
    }
    const uint64_t nsuper = (ntiles + 1023) / 1024;
    hipLaunchKernelGGL(compact_scan_kernel, dim3((uint32_t)nsuper), dim3(64), 0, st, cnt, ntiles, (uint64_t)0,
                       sc.d_tile_off, ntiles, sc.d_super_total);
    // few super tiles (a C3 scan has 64): every scatter wave sums the totals in front of it itself, one launch less
    const bool inline_bases = nsuper <= 512;
    if (!inline_bases)
        hipLaunchKernelGGL(compact_base_kernel, dim3(1), dim3(64), 0, st, (const uint32_t*)sc.d_super_total,
                           sc.d_super_base, nsuper);
    // the scatter grid must also cover the threads that write the per-segment counts (one per segment)
    const uint64_t need = io->counts ? (nseg + kBlock - 1) / kBlock : 0;
    const uint64_t grid = nblocks > need ? nblocks : need;
    hipLaunchKernelGGL(compact_scatter_kernel, dim3((uint32_t)grid), dim3(kBlock), 0, st, *io, seg_len, tps,
                       ntiles, nseg, (const uint32_t*)sc.d_tile_off, (const uint64_t*)sc.d_super_base, (uint64_t)0,
                       inline_bases ? (const uint32_t*)sc.d_super_total : (const uint32_t*)nullptr);
    LRC_HIP(hipGetLastError());
    return LRC_OK;
}


int lrc_compact_dev(lrc_ctx* ctx, uint64_t nseg, uint64_t seg_len, const lrc_compact_io* io,
                    void* stream) {
    if (!ctx || !io) return fail(LRC_ERR_INVALID_ARG, "lrc_compact_dev: NULL argument");
    if (nseg == 0 || seg_len == 0) return LRC_OK;
    if (!io->t) return fail(LRC_ERR_INVALID_ARG, "lrc_compact_dev: t is NULL");
    if (((io->out_point3 || io->out_xyzl || io->out_range_origin) && !io->point3) || (io->out_sem && !io->sem) ||
        (io->out_ins && !io->ins) || (io->out_incident_deg && !io->incident_deg))
        return fail(LRC_ERR_INVALID_ARG, "lrc_compact_dev: an output is requested without its input");
    LRC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const uint64_t tps = (seg_len + 63) / 64;
    const uint64_t ntiles = nseg * tps;
    const uint64_t nblocks = (ntiles + kBlock / 64 - 1) / (kBlock / 64);
    if (nblocks > 0x7FFFFFFFull) return fail(LRC_ERR_INVALID_ARG, "lrc_compact_dev: too many entries");
    int set = -1;
    {
        int rc_scratch = compact_scratch_for(ctx, st, ntiles, &set);
        if (rc_scratch) return rc_scratch;
    }
    {
        int rc_enq = enqueue_compaction(ctx->compact_scratch[set], nseg, seg_len, io, st);
        if (rc_enq) return rc_enq;
    }
    LRC_HIP(hipEventRecord(ctx->compact_done[set], st));
    return LRC_OK;
}

// The scratch set of a compaction on stream `st`: the one that stream used last; otherwise the next set in turn, ordered
// behind whatever another stream may still be doing with it (an event wait; free when that work is long done).
static int compact_scratch_for(lrc_ctx* ctx, hipStream_t st, uint64_t ntiles, int* out_set) {
    int set = -1;
    for (int k = 0; k < lrc_ctx::kCompactSets; ++k)
        if (ctx->compact_used[k] && ctx->compact_stream[k] == st) { set = k; break; }
    if (set < 0) {
        set = ctx->compact_next;
        ctx->compact_next = (ctx->compact_next + 1) % lrc_ctx::kCompactSets;
        if (!ctx->compact_done[set]) LRC_HIP(hipEventCreateWithFlags(&ctx->compact_done[set], hipEventDisableTiming));
        else if (ctx->compact_used[set]) LRC_HIP(hipStreamWaitEvent(st, ctx->compact_done[set], 0));
        ctx->compact_stream[set] = st;
        ctx->compact_used[set] = true;
    }
    lrc_ctx::TileScratch& sc = ctx->compact_scratch[set];
    if (sc.tile_cap < ntiles + 1 && sc.d_tile_off) {
        // growing frees the old arrays: whatever still reads them must be done first
        LRC_HIP(hipEventSynchronize(ctx->compact_done[set]));
    }
    *out_set = set;
    return ensure_tile_scratch(ctx, sc, ntiles);
}

static int ensure_tile_scratch(lrc_ctx* ctx, lrc_ctx::TileScratch& sc, uint64_t ntiles) {
    (void)ctx;
    if (sc.tile_cap >= ntiles + 1) return LRC_OK;
    if (sc.d_tile_off) { (void)hipFree(sc.d_tile_off); sc.d_tile_off = nullptr; }
    if (sc.d_tile_cnt) { (void)hipFree(sc.d_tile_cnt); sc.d_tile_cnt = nullptr; }
    if (sc.d_super_total) { (void)hipFree(sc.d_super_total); sc.d_super_total = nullptr; }
    if (sc.d_super_base) { (void)hipFree(sc.d_super_base); sc.d_super_base = nullptr; }
    sc.tile_cap = 0;
    LRC_HIP(hipMalloc((void**)&sc.d_tile_off, (ntiles + 1) * 4));
    LRC_HIP(hipMalloc((void**)&sc.d_tile_cnt, (ntiles + 1) * 4));
    LRC_HIP(hipMalloc((void**)&sc.d_super_total, ((ntiles + 1023) / 1024 + 1) * 4));
    LRC_HIP(hipMalloc((void**)&sc.d_super_base, ((ntiles + 1023) / 1024 + 1) * 8));
    sc.tile_cap = ntiles + 1;
    return LRC_OK;
}

// ---- the scan pipeline (include/lidarcast.h: lrc_pipe_*) ---------------------------------------------------------------------
// Consecutive pose batches of one scene, scanned and compacted with the launches overlapped INSIDE the library.
// What the measurements of round 4 say about this chip (DESIGN.md "the launch tail"; profiles/r04_chain_timeline_before.txt,
// r04_queue_share.txt, r04_pipe_arrangements.txt):
//   * a trace launch -- 65 536 one-wave workgroups that take a wave slot the moment it is freed -- keeps the dispatcher to
//     itself while it has workgroups left: another stream's kernel gets a trickle of slots if its workgroups are one wave
//     (a 32-workgroup kernel: 30-160 us) and none at all if they are 4 or 16 waves.  When the launch has handed out its
//     last workgroup the next READY kernel of any stream takes over, into the slots the launch's long last waves leave
//     empty (a 64-pose launch alone loses 50-60 us to that tail).
//   * so two trace launches on two streams with NOTHING between two launches of a stream overlap perfectly (+12 %), and
//     the moment a compaction sits between them (trace -> scan -> scatter -> trace per stream) it waits for the OTHER
//     stream's whole dispatch phase, the trace behind it waits too, and the two chains run in lock step (+0...6 %).
// Hence: trace launches of consecutive submits alternate between two streams; the 32-workgroup scan pass of submit k sits
// behind its trace (it trickles in during the other stream's launch, nothing waits for it); and the scatter of submit k is not
// a launch at all: it rides at the FRONT of the trace launch of submit k+2 (same stream, TraceParams::pre: one tile per
// one-wave workgroup), so it is handed out exactly when the previous launch's tail begins and is gone in microseconds.  Four
// record sets: set k is read by launch k+2 and written again by launch k+4, both on its own stream.
struct lrc_pipe {
    lrc_scene* scene = nullptr;
    int device = 0;
    uint64_t max_poses = 0, rays_per_pose = 0, cap = 0;      // cap = max_poses * rays_per_pose records per set
    static constexpr int kSets = 4;
    void* slab[kSets] = {};                                  // one allocation per record set
    lrc_hits rec[kSets] = {};
    lrc_compact_io out[kSets] = {};                          // the caller's output buffers of the submit that used the set
    uint64_t poses[kSets] = {};                              // its pose count
    bool pending[kSets] = {};                                // scanned, scan pass enqueued, rows not yet scattered
    lrc_ctx::TileScratch scratch[2];                         // per trace stream: offsets of the scan waiting for its scatter
    hipStream_t s_trace[2] = {};
    hipEvent_t ev_in[kSets] = {}, ev_t0[kSets] = {}, ev_trace[kSets] = {}, ev_flush[2] = {};
    bool fused = true;                                       // false: N % 64 != 0 or > 512 super tiles: plain chain per stream
    uint64_t ticket = 0;                                     // submits so far; submit k uses set k % 4, trace stream k % 2
    // sharded submits (lrc_pipe_submit_sharded): scratch of the scan over ALL ranks' keep counts, per trace stream, and the
    // direction table transposed for the rebuild (once per table)
    lrc_ctx::TileScratch gscratch[2];
    double* d_dirs_soa = nullptr;
    const double* soa_of = nullptr;
};

int lrc_pipe_destroy(lrc_pipe* pp) {
    if (!pp) return LRC_OK;
    (void)hipSetDevice(pp->device);
    for (hipStream_t st : {pp->s_trace[0], pp->s_trace[1]})
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (int k = 0; k < lrc_pipe::kSets; ++k) {
        if (pp->slab[k]) (void)hipFree(pp->slab[k]);
        for (hipEvent_t e : {pp->ev_in[k], pp->ev_t0[k], pp->ev_trace[k]}) if (e) (void)hipEventDestroy(e);
    }
    for (int k = 0; k < 2; ++k) {
        if (pp->ev_flush[k]) (void)hipEventDestroy(pp->ev_flush[k]);
        for (lrc_ctx::TileScratch* scp : {&pp->scratch[k], &pp->gscratch[k]}) {
            lrc_ctx::TileScratch& sc = *scp;
            if (sc.d_tile_off) (void)hipFree(sc.d_tile_off);
            if (sc.d_tile_cnt) (void)hipFree(sc.d_tile_cnt);
            if (sc.d_super_total) (void)hipFree(sc.d_super_total);
            if (sc.d_super_base) (void)hipFree(sc.d_super_base);
        }
    }
    if (pp->d_dirs_soa) (void)hipFree(pp->d_dirs_soa);
    delete pp;
    return LRC_OK;
}

int lrc_pipe_create(lrc_scene* s, uint64_t max_poses, uint64_t rays_per_pose, lrc_pipe** out_pipe) {
    if (!out_pipe) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_create: out_pipe is NULL");
    *out_pipe = nullptr;
    if (!s) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_create: scene is NULL");
    if (max_poses == 0 || rays_per_pose == 0) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_create: max_poses and rays_per_pose must be positive");
    if (max_poses > 0x7FFFFFFFull || max_poses * rays_per_pose / rays_per_pose != max_poses || max_poses * rays_per_pose > (1ull << 36))
        return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_create: batch too large");
    LRC_HIP(hipSetDevice(s->ctx->device));
    lrc_pipe* pp = new (std::nothrow) lrc_pipe();
    if (!pp) return fail(LRC_ERR_OOM, "lrc_pipe_create: out of host memory");
    pp->scene = s;
    pp->device = s->ctx->device;
    pp->max_poses = max_poses;
    pp->rays_per_pose = rays_per_pose;
    const uint64_t n = pp->cap = max_poses * rays_per_pose;
    const uint64_t tiles = max_poses * ((rays_per_pose + 63) / 64);
    pp->fused = rays_per_pose % 64 == 0 && (tiles + 1023) / 1024 <= 512;
    auto bail = [&](int rc) { (void)lrc_pipe_destroy(pp); return rc; };
    auto up = [](uint64_t b) { return (b + 255) & ~255ull; };
    // a record set: t | prim | normal3 | point3 | sem | ins | tile_count, 36 B per ray + 4 B per 64 rays
    const uint64_t off_t = 0, off_prim = off_t + up(4 * n), off_nrm = off_prim + up(4 * n), off_pt = off_nrm + up(12 * n),
                   off_sem = off_pt + up(12 * n), off_ins = off_sem + up(2 * n), off_tc = off_ins + up(2 * n),
                   bytes = off_tc + up(4 * ((n + 63) / 64));
    for (int k = 0; k < lrc_pipe::kSets; ++k) {
        hipError_t e = hipMalloc(&pp->slab[k], bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return bail(fail(LRC_ERR_OOM, "lrc_pipe_create: out of device memory for the record sets")); }
        char* b = (char*)pp->slab[k];
        lrc_hits& h = pp->rec[k];
        h.t = (float*)(b + off_t); h.prim = (uint32_t*)(b + off_prim); h.normal3 = (float*)(b + off_nrm);
        h.point3 = (float*)(b + off_pt); h.sem = (uint16_t*)(b + off_sem); h.ins = (uint16_t*)(b + off_ins);
        h.tile_count = (uint32_t*)(b + off_tc);
    }
    for (int k = 0; k < 2; ++k) {
        if (hipStreamCreateWithFlags(&pp->s_trace[k], hipStreamNonBlocking) != hipSuccess) return bail(fail(LRC_ERR_HIP, "lrc_pipe_create: stream"));
        if (hipEventCreateWithFlags(&pp->ev_flush[k], hipEventDisableTiming) != hipSuccess) return bail(fail(LRC_ERR_HIP, "lrc_pipe_create: event"));
        int rc = ensure_tile_scratch(s->ctx, pp->scratch[k], tiles);
        if (rc) return bail(rc);
    }
    for (int k = 0; k < lrc_pipe::kSets; ++k) {
        if (hipEventCreateWithFlags(&pp->ev_in[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreate(&pp->ev_t0[k]) != hipSuccess || hipEventCreate(&pp->ev_trace[k]) != hipSuccess)
            return bail(fail(LRC_ERR_HIP, "lrc_pipe_create: event"));
    }
    *out_pipe = pp;
    return LRC_OK;
}

namespace {
// the compaction input of the records in set `set`
lrc_compact_io pipe_io(const lrc_pipe* pp, int set) {
    lrc_compact_io io = pp->out[set];
    const lrc_hits& h = pp->rec[set];
    io.t = h.t; io.point3 = h.point3; io.sem = h.sem; io.ins = h.ins; io.incident_deg = nullptr; io.tile_count = h.tile_count;
    return io;
}
}  // namespace

int lrc_pipe_submit(lrc_pipe* pp, const double* d_poses16, uint64_t P, const double* d_dirs3, double max_range,
                    const lrc_compact_io* d_out, void* stream, uint64_t* out_ticket) {
    if (!pp || !d_out) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_submit: NULL pipe or output");
    if (P == 0) { if (out_ticket) *out_ticket = pp->ticket; return LRC_OK; }
    if (P > pp->max_poses) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_submit: more poses than the pipeline was created for");
    if (!d_poses16 || !d_dirs3) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_submit: poses16 or dirs3 is NULL");
    if (d_out->out_incident_deg) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_submit: the pipeline's records carry no incident angle");
    lrc_scene* s = pp->scene;
    LRC_HIP(hipSetDevice(s->ctx->device));
    const uint64_t k = pp->ticket;
    const int set = (int)(k % lrc_pipe::kSets), lane = (int)(k % 2);
    hipStream_t T = pp->s_trace[lane];
    // inputs (poses, table) and outputs (the caller's rows and counts) are the caller's as of this point of its stream
    LRC_HIP(hipEventRecord(pp->ev_in[set], (hipStream_t)stream));
    LRC_HIP(hipStreamWaitEvent(T, pp->ev_in[set], 0));
    const uint64_t N = pp->rays_per_pose;
    TraceParams p{};
    p.poses16 = d_poses16;
    p.dirs3 = d_dirs3;
    p.rays_per_pose = N;
    p.total = P * N;
    p.has_center = 1;
    p.max_range = max_range;
    p.out = pp->rec[set];
    // the rows of submit k - 2 (this stream's previous scan; its scan pass was enqueued behind its trace) ride in front
    const int prev = (int)((k + lrc_pipe::kSets - 2) % lrc_pipe::kSets);
    lrc_compact_io pio{};
    if (pp->fused && k >= 2 && pp->pending[prev]) {
        pio = pipe_io(pp, prev);
        const uint64_t tps = N / 64, ntiles = pp->poses[prev] * tps;
        const uint64_t need = pio.counts ? (pp->poses[prev] + kTBlock - 1) / kTBlock : 0;
        const uint64_t tile_blocks = (ntiles + kPreTiles - 1) / kPreTiles;
        p.pre.blocks = (uint32_t)(tile_blocks > need ? tile_blocks : need);
        p.pre.rows_only = (pio.out_xyzl && !pio.out_point3 && !pio.out_sem && !pio.out_ins && !pio.out_index && !pio.out_range_origin) ? 1u : 0u;
        p.pre.seg_len = N; p.pre.tps = tps; p.pre.ntiles = ntiles; p.pre.nseg = pp->poses[prev];
        p.pre.tile_off = pp->scratch[lane].d_tile_off;
        p.pre.super_total = pp->scratch[lane].d_super_total;
        p.pre.io = pio;
    }
    LRC_HIP(hipEventRecord(pp->ev_t0[set], T));
    int rc = launch_trace(s, p, 1, T);
    if (rc) return rc;
    LRC_HIP(hipEventRecord(pp->ev_trace[set], T));
    if (p.pre.blocks) pp->pending[prev] = false;
    pp->out[set] = *d_out;
    pp->poses[set] = P;
    if (pp->fused) {
        // the scan pass over this scan's per-wave keep counts: 32-64 one-wave workgroups behind the trace; the rows follow
        // with this stream's next launch (or with lrc_pipe_wait)
        const uint64_t ntiles = P * (N / 64), nsuper = (ntiles + 1023) / 1024;
        // One wave per super tile.  Beside the other stream's running launch a kernel of this stream is handed a wave slot
        // every few microseconds at best, so in a long run four waves (in within 20 us, 80 us of work each) are 0.4 % ahead of
        // 64 (arriving over 300 us: the next launch of this stream waits behind them) -- but a run ENDS with this pass alone on
        // the GPU, where four waves take 80 us and 64 take 10: over blocks of 20 submits the wide pass is 2.4 % faster, over 300
        // submits 0.4 % slower (profiles/r04_pipe_scan_width.txt).  Callers synchronise more often than every 300 batches.
#ifndef LRC_PIPE_SCAN_WAVES
#define LRC_PIPE_SCAN_WAVES 64
#endif
        hipLaunchKernelGGL(compact_scan_kernel, dim3((uint32_t)(nsuper < LRC_PIPE_SCAN_WAVES ? nsuper : LRC_PIPE_SCAN_WAVES)), dim3(64), 0, T, (const uint32_t*)pp->rec[set].tile_count,
                           ntiles, (uint64_t)0, pp->scratch[lane].d_tile_off, ntiles, pp->scratch[lane].d_super_total);
        LRC_HIP(hipGetLastError());
        pp->pending[set] = true;
    } else {
        lrc_compact_io io = pipe_io(pp, set);
        rc = enqueue_compaction(pp->scratch[lane], P, N, &io, T);
        if (rc) return rc;
    }
    pp->ticket = k + 1;
    if (out_ticket) *out_ticket = k + 1;
    return LRC_OK;
}

// ---- the pipeline on N ranks: the trace writes triangle ids + keep counts into the caller's send slab, the assembly of an
// EARLIER scan of all ranks (its gathered slabs) rides in the leading workgroups of this trace launch ----------------------
namespace {
int check_gathered(const lrc_pipe* pp, const lrc_gathered* g, const char* who) {
    auto bad = [&](const char* m) { return fail(LRC_ERR_INVALID_ARG, std::string(who) + ": " + m); };
    if (!g->d_all_poses16 || !g->d_all_prims || !g->d_all_tile_counts || !g->d_out_xyzl) return bad("NULL member of lrc_gathered");
    if (g->poses_per_slab == 0 || g->poses_per_slab > pp->max_poses || g->num_poses_all == 0 || g->num_poses_all % g->poses_per_slab)
        return bad("num_poses_all must be a multiple of poses_per_slab (<= the pipeline's max_poses)");
    if (g->own_slab >= g->num_poses_all / g->poses_per_slab) return bad("own_slab outside the gathered slabs");
    if (g->slab_stride_bytes % 4 || g->slab_stride_bytes < g->poses_per_slab * pp->rays_per_pose * 4) return bad("slab stride smaller than a slab");
    if (g->own_ticket == 0 || g->own_ticket > pp->ticket || pp->ticket - g->own_ticket >= (uint64_t)lrc_pipe::kSets)
        return bad("the own records of that scan are gone (four sets rotate): assemble within three submits");
    const uint64_t ntiles = g->num_poses_all * (pp->rays_per_pose / 64);
    if (ntiles > 0x7FFFFFFFull) return bad("too many entries");
    return LRC_OK;
}

// scan over all ranks' keep counts (two one-wave kernels on `st`), the rebuild's argument block, the own rows' compaction input
// the scan over all ranks' keep counts (two one-wave kernels on `st`) into the scratch set of g->scan_slot; also what a first
// use needs: the plane table, the transposed direction table
int scan_gathered(lrc_pipe* pp, const lrc_gathered* g, const double* d_dirs3, hipStream_t st, int scan_waves) {
    lrc_scene* s = pp->scene;
    lrc_ctx::TileScratch& sc = pp->gscratch[g->scan_slot & 1u];
    const uint64_t N = pp->rays_per_pose, tps = N / 64, ntiles = g->num_poses_all * tps;
    int rc = ensure_tile_scratch(s->ctx, sc, ntiles);
    if (rc) return rc;
    if ((rc = ensure_prim_plane(s, st))) return rc;
    if (pp->soa_of != d_dirs3) {
        if (!pp->d_dirs_soa) LRC_HIP(hipMalloc((void**)&pp->d_dirs_soa, N * 24));
        hipLaunchKernelGGL(dirs_transpose_kernel, dim3((uint32_t)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, d_dirs3, (uint32_t)N,
                           pp->d_dirs_soa);
        LRC_HIP(hipStreamSynchronize(st));          // once per table: the trace streams read it
        pp->soa_of = d_dirs3;
    }
    const uint64_t stride = g->slab_stride_bytes / 4, nsuper = (ntiles + 1023) / 1024;
    const uint64_t grid = scan_waves > 0 && nsuper > (uint64_t)scan_waves ? (uint64_t)scan_waves : nsuper;
    hipLaunchKernelGGL(compact_scan_kernel, dim3((uint32_t)grid), dim3(64), 0, st, g->d_all_tile_counts, g->poses_per_slab * tps, stride,
                       sc.d_tile_off, ntiles, sc.d_super_total);
    hipLaunchKernelGGL(compact_base_kernel, dim3(1), dim3(64), 0, st, (const uint32_t*)sc.d_super_total, sc.d_super_base, nsuper);
    LRC_HIP(hipGetLastError());
    return LRC_OK;
}

// the rebuild's argument block and the own rows' compaction input, for a gathered scan whose counts have been scanned
int prepare_gathered(lrc_pipe* pp, const lrc_gathered* g, RebuildParams* q, lrc_compact_io* own_io, uint64_t* own_tiles,
                     uint64_t* tile_base) {
    lrc_scene* s = pp->scene;
    lrc_ctx::TileScratch& sc = pp->gscratch[g->scan_slot & 1u];
    const uint64_t N = pp->rays_per_pose, tps = N / 64, ntiles = g->num_poses_all * tps;
    if (sc.tile_cap < ntiles + 1 || !pp->d_dirs_soa || !s->d_prim_plane)
        return fail(LRC_ERR_INVALID_ARG, "lrc_pipe: the gathered scan has not been through lrc_pipe_scan_gathered");
    const uint64_t stride = g->slab_stride_bytes / 4;
    q->poses16 = g->d_all_poses16; q->dirs_soa = pp->d_dirs_soa; q->prims = g->d_all_prims;
    q->pps = (uint32_t)g->poses_per_slab; q->stride = stride; q->seg_len = (uint32_t)N; q->tps = (uint32_t)tps;
    q->ntiles = (uint32_t)ntiles; q->nseg = (uint32_t)g->num_poses_all; q->plane = s->d_prim_plane;
    q->num_prims = (uint32_t)s->info.num_triangles; q->tile_off = sc.d_tile_off; q->super_base = sc.d_super_base;
    q->out_xyzl = (float4*)g->d_out_xyzl; q->counts = g->d_counts; q->skip_slab = (uint32_t)g->own_slab;
    const lrc_hits& h = pp->rec[(g->own_ticket - 1) % lrc_pipe::kSets];
    *own_io = lrc_compact_io{};
    own_io->t = h.t; own_io->point3 = h.point3; own_io->sem = h.sem; own_io->ins = h.ins;
    own_io->out_xyzl = g->d_out_xyzl;
    *own_tiles = pp->poses[(g->own_ticket - 1) % lrc_pipe::kSets] * tps;
    *tile_base = g->own_slab * g->poses_per_slab * tps;
    return LRC_OK;
}
}  // namespace

int lrc_pipe_submit_sharded(lrc_pipe* pp, const double* d_poses16, uint64_t P, const double* d_dirs3, double max_range,
                            uint32_t* d_send_prim, uint32_t* d_send_tile_count, const lrc_gathered* assemble, void* stream,
                            uint64_t* out_ticket) {
    if (!pp) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_submit_sharded: pipe is NULL");
    if (!pp->fused) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_submit_sharded: needs rays_per_pose % 64 == 0");
    if (P == 0 || P > pp->max_poses) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_submit_sharded: pose count outside (0, max_poses]");
    if (!d_poses16 || !d_dirs3 || !d_send_prim || !d_send_tile_count)
        return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_submit_sharded: NULL poses, table or send slab");
    lrc_scene* s = pp->scene;
    if (s->opts.range_noise)
        return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_submit_sharded: a scan with range noise cannot be rebuilt from triangle ids");
    LRC_HIP(hipSetDevice(s->ctx->device));
    int rc;
    if (assemble && (rc = check_gathered(pp, assemble, "lrc_pipe_submit_sharded"))) return rc;
    const uint64_t k = pp->ticket;
    const int set = (int)(k % lrc_pipe::kSets), lane = (int)(k % 2);
    hipStream_t T = pp->s_trace[lane];
    // inputs, the send slab and -- for the assembly -- the gathered slabs are the caller's as of this point of its stream
    LRC_HIP(hipEventRecord(pp->ev_in[set], (hipStream_t)stream));
    LRC_HIP(hipStreamWaitEvent(T, pp->ev_in[set], 0));
    const uint64_t N = pp->rays_per_pose;
    TraceParams p{};
    p.poses16 = d_poses16; p.dirs3 = d_dirs3; p.rays_per_pose = N; p.total = P * N; p.has_center = 1; p.max_range = max_range;
    p.out = pp->rec[set];
    p.out.prim = d_send_prim;                  // the 36-byte record stays complete: its id column IS the send slab
    p.out.tile_count = d_send_tile_count;
    lrc_compact_io own_io{};
    if (assemble) {
        uint64_t own_tiles = 0, tile_base = 0;
        if ((rc = prepare_gathered(pp, assemble, &p.pre.rq, &own_io, &own_tiles, &tile_base))) return rc;
        p.pre.sharded = 1;
        p.pre.io = own_io;
        p.pre.ntiles = own_tiles;
        p.pre.seg_len = N; p.pre.tps = N / 64;
        p.pre.tile_off = pp->gscratch[assemble->scan_slot & 1u].d_tile_off;
        p.pre.super_base = pp->gscratch[assemble->scan_slot & 1u].d_super_base;
        p.pre.tile_base = tile_base;
        p.pre.own_blocks = (uint32_t)((own_tiles + kPreTiles - 1) / kPreTiles);
        const uint64_t rb = ((uint64_t)p.pre.rq.ntiles + LRC_REBUILD_R - 1) / LRC_REBUILD_R;
        const uint64_t need = ((uint64_t)p.pre.rq.nseg + kTBlock - 1) / kTBlock;        // threads for the per-pose counts
        uint64_t blocks = p.pre.own_blocks + rb;
        if (blocks < need) blocks = need;
        p.pre.blocks = (uint32_t)blocks;
    }
    LRC_HIP(hipEventRecord(pp->ev_t0[set], T));
    rc = launch_trace(s, p, 1, T);
    if (rc) return rc;
    LRC_HIP(hipEventRecord(pp->ev_trace[set], T));
    pp->out[set] = lrc_compact_io{};
    pp->poses[set] = P;
    pp->pending[set] = false;                  // nothing local to scatter: the rows appear when the gathered scan is assembled
    pp->ticket = k + 1;
    if (out_ticket) *out_ticket = k + 1;
    return LRC_OK;
}

int lrc_pipe_trace_done(lrc_pipe* pp, uint64_t ticket, void* stream) {
    if (!pp) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_trace_done: pipe is NULL");
    if (ticket == 0 || ticket > pp->ticket || pp->ticket - ticket >= (uint64_t)lrc_pipe::kSets)
        return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_trace_done: the events of that submit have been reused");
    LRC_HIP(hipSetDevice(pp->scene->ctx->device));
    LRC_HIP(hipStreamWaitEvent((hipStream_t)stream, pp->ev_trace[(ticket - 1) % lrc_pipe::kSets], 0));
    return LRC_OK;
}

// the scan over the gathered keep counts, on the caller's COMMUNICATION stream right behind the collective: two one-wave kernels
// that trickle in beside the running trace launch and have a whole step before the launch that carries the assembly needs them
// (between two trace launches of one stream they would hold the second back: DESIGN.md section 5.2)
int lrc_pipe_scan_gathered(lrc_pipe* pp, const double* d_dirs3, const lrc_gathered* g, void* stream) {
    if (!pp || !g || !d_dirs3) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_scan_gathered: NULL argument");
    int rc = check_gathered(pp, g, "lrc_pipe_scan_gathered");
    if (rc) return rc;
    LRC_HIP(hipSetDevice(pp->device));
    return scan_gathered(pp, g, d_dirs3, (hipStream_t)stream, LRC_PIPE_SCAN_WAVES * 4);
}

// the assembly of a gathered (and scanned) scan with the plain kernels, on `stream` (the end of a run: no later launch to ride on)
int lrc_pipe_assemble(lrc_pipe* pp, const double* d_dirs3, const lrc_gathered* g, void* stream) {
    if (!pp || !g || !d_dirs3) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_assemble: NULL argument");
    int rc = check_gathered(pp, g, "lrc_pipe_assemble");
    if (rc) return rc;
    LRC_HIP(hipSetDevice(pp->scene->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    // the own records were written by a trace on an internal stream
    LRC_HIP(hipStreamWaitEvent(st, pp->ev_trace[(g->own_ticket - 1) % lrc_pipe::kSets], 0));
    RebuildParams q{};
    lrc_compact_io io{};
    uint64_t own_tiles = 0, tile_base = 0;
    if ((rc = prepare_gathered(pp, g, &q, &io, &own_tiles, &tile_base))) return rc;
    constexpr int kR = LRC_REBUILD_R;
    const uint64_t wblocks = (((uint64_t)q.ntiles + kR - 1) / kR + kBlock / 64 - 1) / (kBlock / 64);
    hipLaunchKernelGGL(prim_scatter_kernel<kR>, dim3((uint32_t)(wblocks ? wblocks : 1)), dim3(kBlock), 0, st, q);
    const uint64_t N = pp->rays_per_pose, tps = N / 64;
    const uint64_t nblocks = (own_tiles + kBlock / 64 - 1) / (kBlock / 64);
    if (nblocks)
        hipLaunchKernelGGL(compact_scatter_kernel, dim3((uint32_t)nblocks), dim3(kBlock), 0, st, io, N, tps, own_tiles, own_tiles / tps,
                           (const uint32_t*)q.tile_off, (const uint64_t*)q.super_base, tile_base, (const uint32_t*)nullptr);
    LRC_HIP(hipGetLastError());
    return LRC_OK;
}

int lrc_pipe_wait(lrc_pipe* pp, void* stream) {
    if (!pp) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_wait: pipe is NULL");
    if (pp->ticket == 0) return LRC_OK;
    LRC_HIP(hipSetDevice(pp->scene->ctx->device));
    // the rows still waiting for a launch to ride on (the last two submits) are scattered by the plain kernel
    for (uint64_t back = 2; back >= 1; --back) {
        if (pp->ticket < back) continue;
        const uint64_t k = pp->ticket - back;
        const int set = (int)(k % lrc_pipe::kSets), lane = (int)(k % 2);
        if (!pp->pending[set]) continue;
        lrc_compact_io io = pipe_io(pp, set);
        const uint64_t N = pp->rays_per_pose, tps = N / 64, ntiles = pp->poses[set] * tps;
        const uint64_t nblocks = (ntiles + kBlock / 64 - 1) / (kBlock / 64);
        const uint64_t need = io.counts ? (pp->poses[set] + kBlock - 1) / kBlock : 0;
        hipLaunchKernelGGL(compact_scatter_kernel, dim3((uint32_t)(nblocks > need ? nblocks : need)), dim3(kBlock), 0, pp->s_trace[lane],
                           io, N, tps, ntiles, pp->poses[set], (const uint32_t*)pp->scratch[lane].d_tile_off,
                           (const uint64_t*)nullptr, (uint64_t)0, (const uint32_t*)pp->scratch[lane].d_super_total);
        LRC_HIP(hipGetLastError());
        pp->pending[set] = false;
    }
    for (int lane = 0; lane < 2; ++lane) {
        LRC_HIP(hipEventRecord(pp->ev_flush[lane], pp->s_trace[lane]));
        LRC_HIP(hipStreamWaitEvent((hipStream_t)stream, pp->ev_flush[lane], 0));
    }
    return LRC_OK;
}

int lrc_pipe_records(lrc_pipe* pp, uint64_t ticket, lrc_hits* out_records) {
    if (!pp || !out_records) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_records: NULL argument");
    if (ticket == 0 || ticket > pp->ticket || pp->ticket - ticket >= (uint64_t)lrc_pipe::kSets)
        return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_records: the records of that submit are gone (four sets rotate)");
    *out_records = pp->rec[(ticket - 1) % lrc_pipe::kSets];
    return LRC_OK;
}

int lrc_pipe_trace_ms(lrc_pipe* pp, uint64_t ticket, float* out_ms) {
    if (!pp || !out_ms) return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_trace_ms: NULL argument");
    if (ticket == 0 || ticket > pp->ticket || pp->ticket - ticket >= (uint64_t)lrc_pipe::kSets)
        return fail(LRC_ERR_INVALID_ARG, "lrc_pipe_trace_ms: the events of that submit have been reused");
    const int set = (int)((ticket - 1) % lrc_pipe::kSets);
    LRC_HIP(hipSetDevice(pp->scene->ctx->device));
    LRC_HIP(hipEventSynchronize(pp->ev_trace[set]));
    LRC_HIP(hipEventElapsedTime(out_ms, pp->ev_t0[set], pp->ev_trace[set]));
    return LRC_OK;
}

int lrc_cloud_from_ranges_dev(lrc_ctx* ctx, const double* d_poses16, uint64_t P, const double* d_dirs3,
                              uint64_t N, const void* d_t_label, float* d_out_xyzl, uint64_t* d_counts,
                              void* stream) {
    if (!ctx) return fail(LRC_ERR_INVALID_ARG, "lrc_cloud_from_ranges_dev: ctx is NULL");
    if (P == 0 || N == 0) return LRC_OK;
    if (!d_poses16 || !d_dirs3 || !d_t_label || !d_out_xyzl)
        return fail(LRC_ERR_INVALID_ARG, "lrc_cloud_from_ranges_dev: NULL argument");
    LRC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const uint64_t tps = (N + 63) / 64, ntiles = P * tps;
    const uint64_t nblocks = (ntiles + kBlock / 64 - 1) / (kBlock / 64);
    if (nblocks > 0x7FFFFFFFull) return fail(LRC_ERR_INVALID_ARG, "lrc_cloud_from_ranges_dev: too many entries");
    lrc_ctx::TileScratch& sc = ctx->cloud_scratch;
    int rc = ensure_tile_scratch(ctx, sc, ntiles);
    if (rc) return rc;
    const uint64_t nsuper = (ntiles + 1023) / 1024;
    hipLaunchKernelGGL(cloud_count_kernel, dim3((uint32_t)nblocks), dim3(kBlock), 0, st, (const uint2*)d_t_label, N,
                       tps, ntiles, sc.d_tile_cnt);
    hipLaunchKernelGGL(compact_scan_kernel, dim3((uint32_t)nsuper), dim3(64), 0, st,
                       (const uint32_t*)sc.d_tile_cnt, ntiles, (uint64_t)0, sc.d_tile_off, ntiles, sc.d_super_total);
    hipLaunchKernelGGL(compact_base_kernel, dim3(1), dim3(64), 0, st, (const uint32_t*)sc.d_super_total,
                       sc.d_super_base, nsuper);
    const uint64_t need = d_counts ? (P + kBlock - 1) / kBlock : 0;
    const uint64_t grid = nblocks > need ? nblocks : need;
    hipLaunchKernelGGL(cloud_scatter_kernel, dim3((uint32_t)grid), dim3(kBlock), 0, st, d_poses16, d_dirs3,
                       (const uint2*)d_t_label, N, tps, ntiles, P, (const uint32_t*)sc.d_tile_off,
                       (const uint64_t*)sc.d_super_base, (float4*)d_out_xyzl, d_counts);
    LRC_HIP(hipGetLastError());
    return LRC_OK;
}

// Checks the arguments of a rebuild from triangle ids, enqueues its preparation on `st` (tile offsets from the senders'
// counts or from a counting pass, transposed direction table) and fills the kernel argument block.
static int prepare_rebuild(lrc_scene* s, const char* who, const double* d_poses16, uint64_t P, const double* d_dirs3,
                           uint64_t N, const uint32_t* d_prim, const uint32_t* d_tile_count, uint64_t poses_per_slab,
                           uint64_t slab_stride_bytes, float* d_out_xyzl, uint64_t* d_counts, hipStream_t st,
                           RebuildParams* q) {
    auto bad = [&](const char* msg) { return fail(LRC_ERR_INVALID_ARG, std::string(who) + ": " + msg); };
    if (!d_poses16 || !d_dirs3 || !d_prim || !d_out_xyzl) return bad("NULL argument");
    if (s->opts.range_noise)
        return bad("a scan with range noise cannot be rebuilt from triangle ids; gather (t,label) pairs "
                   "(lrc_cloud_from_ranges_dev)");
    const uint64_t tps = (N + 63) / 64, ntiles = P * tps;
    if (poses_per_slab == 0 || poses_per_slab >= P) { poses_per_slab = P; slab_stride_bytes = 0; }
    else if (slab_stride_bytes % 4 || slab_stride_bytes < poses_per_slab * N * 4)
        return bad("slab stride smaller than a slab or not a multiple of 4");
    if (d_tile_count && N % 64) return bad("tile counts need rays_per_pose % 64 == 0");
    if (ntiles > 0x7FFFFFFFull || N > 0x7FFFFFFFull) return bad("too many entries");
    lrc_ctx* ctx = s->ctx;
    lrc_ctx::TileScratch& sc = ctx->cloud_scratch;
    int rc = ensure_tile_scratch(ctx, sc, ntiles);
    if (rc) return rc;
    if ((rc = ensure_prim_plane(s, st))) return rc;
    if (sc.dirs_cap < N) {
        if (sc.d_dirs_soa) { (void)hipFree(sc.d_dirs_soa); sc.d_dirs_soa = nullptr; }
        sc.dirs_cap = 0;
        LRC_HIP(hipMalloc((void**)&sc.d_dirs_soa, N * 24));
        sc.dirs_cap = N;
    }
    const uint64_t stride = slab_stride_bytes / 4;
    const uint64_t nsuper = (ntiles + 1023) / 1024;
    if (d_tile_count) {
        // the senders' trace kernels already counted (lrc_hits.tile_count travels in the slab): no counting pass
        hipLaunchKernelGGL(compact_scan_kernel, dim3((uint32_t)nsuper), dim3(64), 0, st, d_tile_count,
                           poses_per_slab * tps, stride, sc.d_tile_off, ntiles, sc.d_super_total);
    } else {
        const uint64_t nblocks = (ntiles + kBlock / 64 - 1) / (kBlock / 64);
        hipLaunchKernelGGL(prim_count_kernel, dim3((uint32_t)nblocks), dim3(kBlock), 0, st, d_prim, poses_per_slab,
                           stride, N, tps, ntiles, sc.d_tile_cnt);
        hipLaunchKernelGGL(compact_scan_kernel, dim3((uint32_t)nsuper), dim3(64), 0, st,
                           (const uint32_t*)sc.d_tile_cnt, ntiles, (uint64_t)0, sc.d_tile_off, ntiles,
                           sc.d_super_total);
    }
    hipLaunchKernelGGL(compact_base_kernel, dim3(1), dim3(64), 0, st, (const uint32_t*)sc.d_super_total,
                       sc.d_super_base, nsuper);
    hipLaunchKernelGGL(dirs_transpose_kernel, dim3((uint32_t)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       d_dirs3, (uint32_t)N, sc.d_dirs_soa);
    q->poses16 = d_poses16;
    q->dirs_soa = sc.d_dirs_soa;
    q->prims = d_prim;
    q->pps = (uint32_t)poses_per_slab;
    q->stride = stride;
    q->seg_len = (uint32_t)N;
    q->tps = (uint32_t)tps;
    q->ntiles = (uint32_t)ntiles;
    q->nseg = (uint32_t)P;
    q->plane = s->d_prim_plane;
    q->num_prims = (uint32_t)s->info.num_triangles;
    q->tile_off = sc.d_tile_off;
    q->super_base = sc.d_super_base;
    q->out_xyzl = (float4*)d_out_xyzl;
    q->counts = d_counts;
    q->skip_slab = 0xFFFFFFFFu;
    return LRC_OK;
}

int lrc_cloud_from_prims_dev(lrc_scene* s, const double* d_poses16, uint64_t P, const double* d_dirs3, uint64_t N,
                             const uint32_t* d_prim, const uint32_t* d_tile_count, uint64_t poses_per_slab,
                             uint64_t slab_stride_bytes, float* d_out_xyzl, uint64_t* d_counts, void* stream) {
    if (!s) return fail(LRC_ERR_INVALID_ARG, "lrc_cloud_from_prims_dev: scene is NULL");
    if (P == 0 || N == 0) return LRC_OK;
    LRC_HIP(hipSetDevice(s->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    RebuildParams q{};
    int rc = prepare_rebuild(s, "lrc_cloud_from_prims_dev", d_poses16, P, d_dirs3, N, d_prim, d_tile_count,
                             poses_per_slab, slab_stride_bytes, d_out_xyzl, d_counts, st, &q);
    if (rc) return rc;
    constexpr int kR = LRC_REBUILD_R;
    const uint64_t wblocks = (((uint64_t)q.ntiles + kR - 1) / kR + kBlock / 64 - 1) / (kBlock / 64);
    hipLaunchKernelGGL(prim_scatter_kernel<kR>, dim3((uint32_t)(wblocks ? wblocks : 1)), dim3(kBlock), 0, st, q);
    LRC_HIP(hipGetLastError());
    return LRC_OK;
}

int lrc_cloud_from_prims_own_dev(lrc_scene* s, const double* d_poses16, uint64_t P, const double* d_dirs3, uint64_t N,
                                 const uint32_t* d_prim, const uint32_t* d_tile_count, uint64_t poses_per_slab,
                                 uint64_t slab_stride_bytes, uint64_t own_slab, const lrc_compact_io* own,
                                 float* d_out_xyzl, uint64_t* d_counts, void* stream) {
    if (!s) return fail(LRC_ERR_INVALID_ARG, "lrc_cloud_from_prims_own_dev: scene is NULL");
    if (P == 0 || N == 0) return LRC_OK;
    if (!own || !own->t || !own->point3)
        return fail(LRC_ERR_INVALID_ARG, "lrc_cloud_from_prims_own_dev: the own records need t and point3");
    if (!d_tile_count || N % 64 || poses_per_slab == 0 || poses_per_slab > P || own_slab * poses_per_slab >= P)
        return fail(LRC_ERR_INVALID_ARG, "lrc_cloud_from_prims_own_dev: needs per-wave keep counts (rays_per_pose % 64 == 0), "
                                         "slabs and an own slab inside the scan");
    LRC_HIP(hipSetDevice(s->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    RebuildParams q{};
    int rc = prepare_rebuild(s, "lrc_cloud_from_prims_own_dev", d_poses16, P, d_dirs3, N, d_prim, d_tile_count,
                             poses_per_slab, slab_stride_bytes, d_out_xyzl, d_counts, st, &q);
    if (rc) return rc;
    q.skip_slab = (uint32_t)own_slab;
    constexpr int kR = LRC_REBUILD_R;
    const uint64_t wblocks = (((uint64_t)q.ntiles + kR - 1) / kR + kBlock / 64 - 1) / (kBlock / 64);
    hipLaunchKernelGGL(prim_scatter_kernel<kR>, dim3((uint32_t)(wblocks ? wblocks : 1)), dim3(kBlock), 0, st, q);
    // the own poses: rows straight from the local record (what the trace wrote), at the offsets of the assembled cloud
    const uint64_t first_pose = own_slab * poses_per_slab;
    const uint64_t own_poses = first_pose + poses_per_slab <= P ? poses_per_slab : P - first_pose;
    const uint64_t tps = N / 64, own_tiles = own_poses * tps;
    lrc_compact_io io = *own;
    io.counts = nullptr;                 // the per-pose counts of ALL poses come from the rebuild's scan
    io.out_point3 = nullptr; io.out_sem = nullptr; io.out_ins = nullptr; io.out_incident_deg = nullptr;
    io.out_index = nullptr; io.out_range_origin = nullptr;
    io.out_xyzl = d_out_xyzl;
    const uint64_t nblocks = (own_tiles + kBlock / 64 - 1) / (kBlock / 64);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3((uint32_t)nblocks), dim3(kBlock), 0, st, io, N, tps, own_tiles, own_poses,
                       (const uint32_t*)q.tile_off, (const uint64_t*)q.super_base, first_pose * tps, (const uint32_t*)nullptr);
    LRC_HIP(hipGetLastError());
    return LRC_OK;
}

int lrc_cloud_range_stats_dev(lrc_ctx* ctx, const float* d_xyzl, const uint64_t* d_counts, uint64_t num_poses,
                              uint64_t max_rows, float* d_range, float* d_mean, float* d_std, void* stream) {
    if (!ctx) return fail(LRC_ERR_INVALID_ARG, "lrc_cloud_range_stats_dev: ctx is NULL");
    if (num_poses == 0) return LRC_OK;
    if (!d_xyzl || !d_counts || !d_range || !d_mean || !d_std)
        return fail(LRC_ERR_INVALID_ARG, "lrc_cloud_range_stats_dev: NULL argument");
    LRC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    if (max_rows)
        hipLaunchKernelGGL(rows_range_kernel, dim3((uint32_t)((max_rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           (const float4*)d_xyzl, max_rows, d_range);
    const uint64_t need = segment_stats_scratch_values(num_poses, max_rows);
    if (ctx->stat_scratch_cap < need) {
        if (ctx->stat_scratch) { LRC_HIP(hipDeviceSynchronize()); (void)hipFree(ctx->stat_scratch); ctx->stat_scratch = nullptr; ctx->stat_scratch_cap = 0; }
        LRC_HIP(hipMalloc((void**)&ctx->stat_scratch, (need + need / 4) * 4));
        ctx->stat_scratch_cap = need + need / 4;
    }
    launch_segment_stats<float>(st, (const float*)d_range, d_counts, (uint64_t)0, num_poses, ctx->stat_scratch, d_mean, d_std);
    LRC_HIP(hipGetLastError());
    return LRC_OK;
}

int lrc_compact(lrc_ctx* ctx, uint64_t nseg, uint64_t seg_len, const lrc_compact_io* io,
                uint64_t* out_total) {
    if (!ctx || !io) return fail(LRC_ERR_INVALID_ARG, "lrc_compact: NULL argument");
    if (out_total) *out_total = 0;
    const uint64_t n = nseg * seg_len;
    if (!n) return LRC_OK;
    if (!io->t) return fail(LRC_ERR_INVALID_ARG, "lrc_compact: t is NULL");
    LRC_HIP(hipSetDevice(ctx->device));
    DevBuf t, p3, sem, ins, inc, cnt, op3, osem, oins, oinc, oidx, oxyzl;
    lrc_compact_io d{};
    auto up = [&](DevBuf& b, const void* src, size_t bytes, const void** dst) -> int {
        LRC_HIP(hipMalloc(&b.p, bytes));
        LRC_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        *dst = b.p;
        return LRC_OK;
    };
    int rc;
    if ((rc = up(t, io->t, n * 4, (const void**)&d.t))) return rc;
    if (io->point3 && (rc = up(p3, io->point3, n * 12, (const void**)&d.point3))) return rc;
    if (io->sem && (rc = up(sem, io->sem, n * 2, (const void**)&d.sem))) return rc;
    if (io->ins && (rc = up(ins, io->ins, n * 2, (const void**)&d.ins))) return rc;
    if (io->incident_deg && (rc = up(inc, io->incident_deg, n * 8, (const void**)&d.incident_deg))) return rc;
    LRC_HIP(hipMalloc(&cnt.p, nseg * 8)); d.counts = (uint64_t*)cnt.p;
    if (io->out_point3) { LRC_HIP(hipMalloc(&op3.p, n * 12)); d.out_point3 = (float*)op3.p; }
    if (io->out_sem) { LRC_HIP(hipMalloc(&osem.p, n * 2)); d.out_sem = (uint16_t*)osem.p; }
    if (io->out_ins) { LRC_HIP(hipMalloc(&oins.p, n * 2)); d.out_ins = (uint16_t*)oins.p; }
    if (io->out_incident_deg) { LRC_HIP(hipMalloc(&oinc.p, n * 8)); d.out_incident_deg = (double*)oinc.p; }
    if (io->out_index) { LRC_HIP(hipMalloc(&oidx.p, n * 4)); d.out_index = (uint32_t*)oidx.p; }
    if (io->out_xyzl) { LRC_HIP(hipMalloc(&oxyzl.p, n * 16)); d.out_xyzl = (float*)oxyzl.p; }
    if ((rc = lrc_compact_dev(ctx, nseg, seg_len, &d, nullptr))) return rc;
    LRC_HIP(hipDeviceSynchronize());
    std::vector<uint64_t> counts(nseg);
    LRC_HIP(hipMemcpy(counts.data(), d.counts, nseg * 8, hipMemcpyDeviceToHost));
    uint64_t K = 0;
    for (uint64_t c : counts) K += c;
    if (io->counts) std::memcpy(io->counts, counts.data(), nseg * 8);
    if (out_total) *out_total = K;
    if (K) {
        if (io->out_point3) LRC_HIP(hipMemcpy(io->out_point3, d.out_point3, K * 12, hipMemcpyDeviceToHost));
        if (io->out_sem) LRC_HIP(hipMemcpy(io->out_sem, d.out_sem, K * 2, hipMemcpyDeviceToHost));
        if (io->out_ins) LRC_HIP(hipMemcpy(io->out_ins, d.out_ins, K * 2, hipMemcpyDeviceToHost));
        if (io->out_incident_deg)
            LRC_HIP(hipMemcpy(io->out_incident_deg, d.out_incident_deg, K * 8, hipMemcpyDeviceToHost));
        if (io->out_index) LRC_HIP(hipMemcpy(io->out_index, d.out_index, K * 4, hipMemcpyDeviceToHost));
        if (io->out_xyzl) LRC_HIP(hipMemcpy(io->out_xyzl, d.out_xyzl, K * 16, hipMemcpyDeviceToHost));
    }
    return LRC_OK;
}


// ---- scan straight to the reference's variable-length frames ----------------------------------------------------
namespace {
// The *_compact entry points enqueue their input copies (on the context's compute stream, which consumes them) before
// anything can fail: an error return must not leave such a copy in flight over staging buffers the next call reuses.
struct SyncUnlessOk {
    bool armed = true;
    ~SyncUnlessOk() { if (armed) (void)hipDeviceSynchronize(); }
    int done(int rc) { if (rc == LRC_OK) armed = false; return rc; }
};

// compacted frame arrays in HBM (context pool) and the record set they are compacted from
struct FrameStage {
    DevBuf t, p3, sem, ins, inc, tile;                     // fixed-stride records
    DevBuf cnt, op3, osem, oins, oinc, oidx, oxyzl, orng;   // compacted outputs
    DevBuf fstat;                                           // per-pose statistics: 4 x P doubles
    lrc_hits rec{};
    lrc_compact_io io{};
    double* d_stats = nullptr;
    uint64_t stat_partial = 0;                                // values of chunk-sum scratch per column, behind the 4 P results
    int alloc(lrc_ctx* ctx, const lrc_frames& f, uint64_t P, uint64_t n) {
        int rc;
        const bool rstats = f.range_origin_mean || f.range_origin_std, istats = f.incident_mean || f.incident_std;
        const bool want_pt = f.point3 || f.xyzl || f.range_origin || rstats;
        const bool want_sem = f.sem || f.xyzl, want_ins = f.ins || f.xyzl;
        if ((rc = t.get(ctx, kPoolT, n * 4))) return rc;
        rec.t = (float*)t.p;
        if (want_pt) { if ((rc = p3.get(ctx, kPoolPoint, n * 12))) return rc; rec.point3 = (float*)p3.p; }
        if (want_sem) { if ((rc = sem.get(ctx, kPoolSem, n * 2))) return rc; rec.sem = (uint16_t*)sem.p; }
        if (want_ins) { if ((rc = ins.get(ctx, kPoolIns, n * 2))) return rc; rec.ins = (uint16_t*)ins.p; }
        if (f.incident_deg || istats) { if ((rc = inc.get(ctx, kPoolInc, n * 8))) return rc; rec.incident_deg = (double*)inc.p; }
        if (rstats || istats) {
            // 4 x P doubles of results, then the chunk sums of the two columns (segment_chunk_sums_kernel)
            stat_partial = segment_stats_scratch_values(P, n);
            if ((rc = fstat.get(ctx, kPoolFrameStats, (P * 4 + 2 * stat_partial) * 8))) return rc;
            d_stats = (double*)fstat.p;
        }
        if ((rc = tile.get(ctx, kPoolTile, ((n + 63) / 64 + 1) * 4))) return rc;
        rec.tile_count = (uint32_t*)tile.p;
        io.t = rec.t; io.point3 = rec.point3; io.sem = rec.sem; io.ins = rec.ins; io.incident_deg = rec.incident_deg;
        io.tile_count = rec.tile_count;
        if ((rc = cnt.get(ctx, kPoolCounts, P * 8))) return rc;
        io.counts = (uint64_t*)cnt.p;
        if (f.point3) { if ((rc = op3.get(ctx, kPoolOutPoint, n * 12))) return rc; io.out_point3 = (float*)op3.p; }
        if (f.sem) { if ((rc = osem.get(ctx, kPoolOutSem, n * 2))) return rc; io.out_sem = (uint16_t*)osem.p; }
        if (f.ins) { if ((rc = oins.get(ctx, kPoolOutIns, n * 2))) return rc; io.out_ins = (uint16_t*)oins.p; }
        if (f.incident_deg || istats) { if ((rc = oinc.get(ctx, kPoolOutInc, n * 8))) return rc; io.out_incident_deg = (double*)oinc.p; }
        if (f.index) { if ((rc = oidx.get(ctx, kPoolOutIdx, n * 4))) return rc; io.out_index = (uint32_t*)oidx.p; }
        if (f.xyzl) { if ((rc = oxyzl.get(ctx, kPoolOutXyzl, n * 16))) return rc; io.out_xyzl = (float*)oxyzl.p; }
        if (f.range_origin || rstats) { if ((rc = orng.get(ctx, kPoolOutRange, n * 4))) return rc; io.out_range_origin = (float*)orng.p; }
        return LRC_OK;
    }
};

int ensure_streams(lrc_ctx* ctx) {
    if (!ctx->s_compute) LRC_HIP(hipStreamCreateWithFlags(&ctx->s_compute, hipStreamNonBlocking));
    if (!ctx->s_copy) LRC_HIP(hipStreamCreateWithFlags(&ctx->s_copy, hipStreamNonBlocking));
    if (!ctx->s_stats) LRC_HIP(hipStreamCreateWithFlags(&ctx->s_stats, hipStreamNonBlocking));
    for (hipEvent_t& e : ctx->ev_chunk)
        if (!e) LRC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t& e : ctx->ev_compact)
        if (!e) LRC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return LRC_OK;
}

// Trace -> compaction -> kept rows to the host, pipelined over contiguous chunks of poses: while chunk c's rows cross
// PCIe on the copy stream, chunk c+1 is traced and compacted on the compute stream.  Each chunk compacts into its own
// worst-case region of the device arrays (rows [p0*N, ...)); the host destination offset is the running row count, so
// the host arrays come out packed in pose order (np.vstack order) without a second pass.  Pinned destinations
// (lrc_host_alloc) make the transfers true DMA.  `p` describes the whole scan (inputs already in HBM).
int frames_finish(lrc_scene* s, TraceParams& p, int gen, FrameStage& st, uint64_t P, uint64_t N, const lrc_frames* out,
                  uint64_t capacity, uint64_t* out_total) {
    lrc_ctx* ctx = s->ctx;
    int rc = ensure_streams(ctx);
    if (rc) return rc;
    // Every exit -- the error returns below too -- leaves the three streams idle: the caller's stage buffers go back to
    // the context pool and its pinned destinations to Python as soon as this function returns.
    struct Drain {
        lrc_ctx* c;
        ~Drain() {
            (void)hipStreamSynchronize(c->s_compute);
            (void)hipStreamSynchronize(c->s_copy);
            (void)hipStreamSynchronize(c->s_stats);
        }
    } drain{ctx};
    if (ctx->h_counts_cap < P) {
        if (ctx->h_counts) { (void)hipHostFree(ctx->h_counts); ctx->h_counts = nullptr; ctx->h_counts_cap = 0; }
        LRC_HIP(hipHostMalloc((void**)&ctx->h_counts, (P + P / 4 + 64) * 8 * 5, hipHostMallocDefault));
        ctx->h_counts_cap = P + P / 4 + 64;
    }
    // chunks only pay when there is enough to overlap, and need pose boundaries on 64-ray tiles (fused keep counts)
    uint64_t chunks = (P * N >= (1u << 20) && N % 64 == 0) ? (P < 4 ? P : 4) : 1;
    // Graded chunks: an eighth of the poses, then up to the half, then the rest.  The transfer of the rows is the longest
    // stage (67 MB at 50-56 GB/s for C3): it starts as soon as a few poses are traced, and the later, larger chunks are ready
    // before the copy engine asks for them (C3: 1.48 ms against 1.53 for four equal chunks; tools/chunk_scheme_sweep.sh,
    // profiles/r03_chunk_schemes.txt).  Equal chunks for short trajectories.
    bool graded = chunks == 4 && P >= 8;
    uint32_t ends32[8] = {4, 16, 32, 32, 32, 32, 32, 32};       // chunk ends in 32nds of the trajectory
    if (graded) chunks = 3;
#ifdef LRC_VARIANTS
    {   // A/B knob: LRC_CHUNK_SCHEME = "a,b,...,32" chunk ends in 32nds; "0" = four equal chunks
        const char* e = std::getenv("LRC_CHUNK_SCHEME");
        if (e && graded) {
            if (e[0] == '0') { graded = false; chunks = 4; }
            else {
                uint64_t k = 0;
                for (const char* q = e; *q && k < 8; ++k) { ends32[k] = (uint32_t)std::strtoul(q, (char**)&q, 10); if (*q == ',') ++q; }
                if (k && ends32[k - 1] == 32) chunks = k;
            }
        }
    }
#endif
    auto chunk_end = [&](uint64_t c) -> uint64_t {
        if (!graded) return P * (c + 1) / chunks;
        return c + 1 == chunks ? P : P * ends32[c] / 32;
    };
    const float* noise = p.range_noise ? p.range_noise : s->opts.range_noise;   // staged in HBM by NoiseStage
    // (the callers enqueue their input copies on the compute stream: nothing to wait for here)
    const uint64_t n_all = P * N;
    const size_t row_bytes = (out->point3 ? 12 : 0) + (out->sem ? 2 : 0) + (out->ins ? 2 : 0) + (out->incident_deg ? 8 : 0) +
                             (out->index ? 4 : 0) + (out->xyzl ? 16 : 0) + (out->range_origin ? 4 : 0);
    if (chunks == 1 && capacity >= n_all && n_all * row_bytes <= (4u << 20)) {
        // A small call -- the reference's own loop asks for ONE pose per call (s3dis_simulator.py:254-264): everything on
        // one stream and ONE synchronisation.  The rows are copied at their worst-case length together with the counts
        // (a pose's rows are ~1 MB: cheaper than a second round trip to learn the exact length first).
        hipStream_t cs = ctx->s_compute;
        TraceParams q = p;
        q.out = st.rec;
        lrc_scan_options saved = s->opts;
        q.range_noise = nullptr;
        if (noise) { s->opts.range_noise = noise; s->opts.range_noise_len = q.total; }
        rc = launch_trace(s, q, gen, cs);
        s->opts = saved;
        if (rc) return rc;
        if ((rc = lrc_compact_dev(ctx, P, N, &st.io, cs))) return rc;
        LRC_HIP(hipMemcpyAsync(ctx->h_counts, st.io.counts, P * 8, hipMemcpyDeviceToHost, cs));
        double* hs = (double*)(ctx->h_counts + ctx->h_counts_cap);
        if (st.d_stats) {
            float* rm = (float*)st.d_stats;            float* rs = (float*)(st.d_stats + P);
            double* im = st.d_stats + 2 * P;           double* is = st.d_stats + 3 * P;
            if (out->range_origin_mean || out->range_origin_std) {
                launch_segment_stats<float>(cs, (const float*)st.io.out_range_origin, (const uint64_t*)st.io.counts, (uint64_t)0, P,
                                            (float*)(st.d_stats + 4 * P), rm, rs);
                LRC_HIP(hipMemcpyAsync((float*)hs, rm, P * 4, hipMemcpyDeviceToHost, cs));
                LRC_HIP(hipMemcpyAsync((float*)(hs + ctx->h_counts_cap), rs, P * 4, hipMemcpyDeviceToHost, cs));
            }
            if (out->incident_mean || out->incident_std) {
                launch_segment_stats<double>(cs, (const double*)st.io.out_incident_deg, (const uint64_t*)st.io.counts, (uint64_t)0, P,
                                             st.d_stats + 4 * P + st.stat_partial, im, is);
                LRC_HIP(hipMemcpyAsync(hs + 2 * ctx->h_counts_cap, im, P * 8, hipMemcpyDeviceToHost, cs));
                LRC_HIP(hipMemcpyAsync(hs + 3 * ctx->h_counts_cap, is, P * 8, hipMemcpyDeviceToHost, cs));
            }
            LRC_HIP(hipGetLastError());
        }
        if (out->point3) LRC_HIP(hipMemcpyAsync(out->point3, st.io.out_point3, n_all * 12, hipMemcpyDeviceToHost, cs));
        if (out->sem) LRC_HIP(hipMemcpyAsync(out->sem, st.io.out_sem, n_all * 2, hipMemcpyDeviceToHost, cs));
        if (out->ins) LRC_HIP(hipMemcpyAsync(out->ins, st.io.out_ins, n_all * 2, hipMemcpyDeviceToHost, cs));
        if (out->incident_deg) LRC_HIP(hipMemcpyAsync(out->incident_deg, st.io.out_incident_deg, n_all * 8, hipMemcpyDeviceToHost, cs));
        if (out->index) LRC_HIP(hipMemcpyAsync(out->index, st.io.out_index, n_all * 4, hipMemcpyDeviceToHost, cs));
        if (out->xyzl) LRC_HIP(hipMemcpyAsync(out->xyzl, st.io.out_xyzl, n_all * 16, hipMemcpyDeviceToHost, cs));
        if (out->range_origin) LRC_HIP(hipMemcpyAsync(out->range_origin, st.io.out_range_origin, n_all * 4, hipMemcpyDeviceToHost, cs));
        LRC_HIP(hipStreamSynchronize(cs));
        uint64_t K1 = 0;
        for (uint64_t k = 0; k < P; ++k) {
            out->counts[k] = ctx->h_counts[k];
            K1 += ctx->h_counts[k];
            if (out->range_origin_mean) out->range_origin_mean[k] = ((const float*)hs)[k];
            if (out->range_origin_std) out->range_origin_std[k] = ((const float*)(hs + ctx->h_counts_cap))[k];
            if (out->incident_mean) out->incident_mean[k] = hs[2 * ctx->h_counts_cap + k];
            if (out->incident_std) out->incident_std[k] = hs[3 * ctx->h_counts_cap + k];
        }
        if (out_total) *out_total = K1;
        return LRC_OK;
    }
    uint64_t p0 = 0;
    for (uint64_t c = 0; c < chunks; ++c) {
        const uint64_t p1 = chunk_end(c), np_ = p1 - p0, r0 = p0 * N;
        TraceParams q = p;
        q.poses16 = p.poses16 + p0 * 16;
        if (q.angles2) q.angles2 = p.angles2 + r0 * 2;
        if (q.rays6) q.rays6 = p.rays6 + r0 * 6;
        if (q.seg_centers3) q.seg_centers3 = p.seg_centers3 + p0 * 3;
        if (q.keep_mask) q.keep_mask = p.keep_mask + r0;
        q.total = np_ * N;
        q.range_noise = nullptr;
        lrc_scan_options saved = s->opts;
        if (noise) { s->opts.range_noise = noise + r0; s->opts.range_noise_len = q.total; }
        q.out = st.rec;
        q.out.t = st.rec.t + r0;
        if (q.out.point3) q.out.point3 = st.rec.point3 + r0 * 3;
        if (q.out.sem) q.out.sem = st.rec.sem + r0;
        if (q.out.ins) q.out.ins = st.rec.ins + r0;
        if (q.out.incident_deg) q.out.incident_deg = st.rec.incident_deg + r0;
        q.out.tile_count = st.rec.tile_count + (r0 + 63) / 64;
        rc = launch_trace(s, q, gen, ctx->s_compute);
        s->opts = saved;
        if (rc) return rc;
        lrc_compact_io io = st.io;
        io.t = q.out.t; io.point3 = q.out.point3; io.sem = q.out.sem; io.ins = q.out.ins;
        io.incident_deg = q.out.incident_deg; io.tile_count = q.out.tile_count;
        io.counts = st.io.counts + p0;
        if (io.out_point3) io.out_point3 = st.io.out_point3 + r0 * 3;
        if (io.out_sem) io.out_sem = st.io.out_sem + r0;
        if (io.out_ins) io.out_ins = st.io.out_ins + r0;
        if (io.out_incident_deg) io.out_incident_deg = st.io.out_incident_deg + r0;
        if (io.out_index) io.out_index = st.io.out_index + r0;
        if (io.out_xyzl) io.out_xyzl = st.io.out_xyzl + r0 * 4;
        if (io.out_range_origin) io.out_range_origin = st.io.out_range_origin + r0;
        if ((rc = lrc_compact_dev(ctx, np_, N, &io, ctx->s_compute))) return rc;
        LRC_HIP(hipMemcpyAsync(ctx->h_counts + p0, st.io.counts + p0, np_ * 8, hipMemcpyDeviceToHost, ctx->s_compute));
        LRC_HIP(hipEventRecord(ctx->ev_chunk[c], ctx->s_compute));
        if (st.d_stats) {
            // per-pose statistics of the compacted columns, numpy's arithmetic (lrc_stats.h), on a stream of their own:
            // they need this chunk's compaction and nothing else, so they run beside the next chunk's trace and the
            // row transfers.  Floats in the first two blocks of P doubles' worth of space, doubles in the last two.
            hipStream_t ss = ctx->s_stats;
            LRC_HIP(hipStreamWaitEvent(ss, ctx->ev_chunk[c], 0));
            float* rm = (float*)st.d_stats;            float* rs = (float*)(st.d_stats + P);
            double* im = st.d_stats + 2 * P;           double* is = st.d_stats + 3 * P;
            double* hs = (double*)(ctx->h_counts + ctx->h_counts_cap);
            if (out->range_origin_mean || out->range_origin_std) {
                launch_segment_stats<float>(ss, (const float*)st.io.out_range_origin, (const uint64_t*)(st.io.counts + p0), r0, np_,
                                            (float*)(st.d_stats + 4 * P) + r0 / 8192 + p0, rm + p0, rs + p0);
                LRC_HIP(hipMemcpyAsync((float*)hs + p0, rm + p0, np_ * 4, hipMemcpyDeviceToHost, ss));
                LRC_HIP(hipMemcpyAsync((float*)(hs + ctx->h_counts_cap) + p0, rs + p0, np_ * 4, hipMemcpyDeviceToHost, ss));
            }
            if (out->incident_mean || out->incident_std) {
                launch_segment_stats<double>(ss, (const double*)st.io.out_incident_deg, (const uint64_t*)(st.io.counts + p0), r0, np_,
                                             st.d_stats + 4 * P + st.stat_partial + r0 / 8192 + p0, im + p0, is + p0);
                LRC_HIP(hipMemcpyAsync(hs + 2 * ctx->h_counts_cap + p0, im + p0, np_ * 8, hipMemcpyDeviceToHost, ss));
                LRC_HIP(hipMemcpyAsync(hs + 3 * ctx->h_counts_cap + p0, is + p0, np_ * 8, hipMemcpyDeviceToHost, ss));
            }
            LRC_HIP(hipGetLastError());
        }
        p0 = p1;
    }
    uint64_t K = 0;
    int status = LRC_OK;
    p0 = 0;
    for (uint64_t c = 0; c < chunks; ++c) {
        const uint64_t p1 = chunk_end(c), r0 = p0 * N;
        LRC_HIP(hipEventSynchronize(ctx->ev_chunk[c]));
        uint64_t Kc = 0;
        for (uint64_t k = p0; k < p1; ++k) { out->counts[k] = ctx->h_counts[k]; Kc += ctx->h_counts[k]; }
        if (K + Kc > capacity) status = LRC_ERR_INVALID_ARG;        // keep counting: the caller learns the size needed
        if (status == LRC_OK && Kc) {
            hipStream_t cs = ctx->s_copy;
            if (out->point3) LRC_HIP(hipMemcpyAsync(out->point3 + K * 3, st.io.out_point3 + r0 * 3, Kc * 12, hipMemcpyDeviceToHost, cs));
            if (out->sem) LRC_HIP(hipMemcpyAsync(out->sem + K, st.io.out_sem + r0, Kc * 2, hipMemcpyDeviceToHost, cs));
            if (out->ins) LRC_HIP(hipMemcpyAsync(out->ins + K, st.io.out_ins + r0, Kc * 2, hipMemcpyDeviceToHost, cs));
            if (out->incident_deg)
                LRC_HIP(hipMemcpyAsync(out->incident_deg + K, st.io.out_incident_deg + r0, Kc * 8, hipMemcpyDeviceToHost, cs));
            if (out->index) LRC_HIP(hipMemcpyAsync(out->index + K, st.io.out_index + r0, Kc * 4, hipMemcpyDeviceToHost, cs));
            if (out->xyzl) LRC_HIP(hipMemcpyAsync(out->xyzl + K * 4, st.io.out_xyzl + r0 * 4, Kc * 16, hipMemcpyDeviceToHost, cs));
            if (out->range_origin)
                LRC_HIP(hipMemcpyAsync(out->range_origin + K, st.io.out_range_origin + r0, Kc * 4, hipMemcpyDeviceToHost, cs));
        }
        K += Kc;
        p0 = p1;
    }
    LRC_HIP(hipStreamSynchronize(ctx->s_copy));
    LRC_HIP(hipStreamSynchronize(ctx->s_compute));
    if (st.d_stats) {
        LRC_HIP(hipStreamSynchronize(ctx->s_stats));
        const double* hs = (const double*)(ctx->h_counts + ctx->h_counts_cap);
        for (uint64_t k = 0; k < P; ++k) {
            if (out->range_origin_mean) out->range_origin_mean[k] = ((const float*)hs)[k];
            if (out->range_origin_std) out->range_origin_std[k] = ((const float*)(hs + ctx->h_counts_cap))[k];
            if (out->incident_mean) out->incident_mean[k] = hs[2 * ctx->h_counts_cap + k];
            if (out->incident_std) out->incident_std[k] = hs[3 * ctx->h_counts_cap + k];
        }
    }
    if (out_total) *out_total = K;
    if (status != LRC_OK)
        return fail(LRC_ERR_INVALID_ARG, "frame buffers too small: capacity " + std::to_string(capacity) +
                                             " rows, the scan kept " + std::to_string(K));
    return LRC_OK;
}
}  // namespace

int lrc_host_alloc(lrc_ctx* ctx, uint64_t bytes, void** out_ptr) {
    if (!ctx || !out_ptr) return fail(LRC_ERR_INVALID_ARG, "lrc_host_alloc: NULL argument");
    *out_ptr = nullptr;
    if (!bytes) return LRC_OK;
    LRC_HIP(hipSetDevice(ctx->device));
    LRC_HIP(hipHostMalloc(out_ptr, bytes, hipHostMallocDefault));
    return LRC_OK;
}

int lrc_host_free(lrc_ctx* ctx, void* ptr) {
    if (!ptr) return LRC_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    LRC_HIP(hipHostFree(ptr));
    return LRC_OK;
}

static int scan_compact_impl(lrc_scene* s, const double* poses16, uint64_t P, const double* dirs3, uint64_t N,
                             const lrc_grid* grid, double max_range, const lrc_frames* out, uint64_t capacity,
                             uint64_t* out_total, const double* d_dirs3 = nullptr);

int lrc_table_create(lrc_ctx* ctx, const double* dirs3, uint64_t N, lrc_table** out_table) {
    if (!out_table) return fail(LRC_ERR_INVALID_ARG, "lrc_table_create: out_table is NULL");
    *out_table = nullptr;
    if (!ctx || !dirs3 || !N) return fail(LRC_ERR_INVALID_ARG, "lrc_table_create: NULL context / table or empty table");
    LRC_HIP(hipSetDevice(ctx->device));
    lrc_table* t = new (std::nothrow) lrc_table();
    if (!t) return fail(LRC_ERR_OOM, "lrc_table_create: out of host memory");
    t->ctx = ctx;
    t->n = N;
    hipError_t e = hipMalloc((void**)&t->d_dirs3, N * 24);
    if (e == hipSuccess) e = hipMemcpy(t->d_dirs3, dirs3, N * 24, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (t->d_dirs3) (void)hipFree(t->d_dirs3);
        delete t;
        return fail(e == hipErrorOutOfMemory ? LRC_ERR_OOM : LRC_ERR_HIP, std::string("lrc_table_create: ") + hipGetErrorString(e));
    }
    *out_table = t;
    return LRC_OK;
}

int lrc_table_destroy(lrc_table* t) {
    if (!t) return LRC_OK;
    if (t->ctx) (void)hipSetDevice(t->ctx->device);
    if (t->d_dirs3) (void)hipFree(t->d_dirs3);
    delete t;
    return LRC_OK;
}

int lrc_scan_table_compact(lrc_scene* s, const double* poses16, uint64_t P, const lrc_table* table, const lrc_grid* grid,
                           double max_range, const lrc_frames* out, uint64_t capacity, uint64_t* out_total) {
    if (out_total) *out_total = 0;
    if (!table) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_table_compact: table is NULL");
    if (s && table->ctx != s->ctx) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_table_compact: table and scene belong to different contexts");
    if (grid) {
        int rc = check_grid("lrc_scan_table_compact", grid, table->n);
        if (rc) return rc;
    }
    return scan_compact_impl(s, poses16, P, nullptr, table->n, grid, max_range, out, capacity, out_total, table->d_dirs3);
}

int lrc_scan_poses_compact(lrc_scene* s, const double* poses16, uint64_t P, const double* dirs3, uint64_t N,
                           double max_range, const lrc_frames* out, uint64_t capacity, uint64_t* out_total) {
    return scan_compact_impl(s, poses16, P, dirs3, N, nullptr, max_range, out, capacity, out_total);
}

int lrc_scan_grid_compact(lrc_scene* s, const double* poses16, uint64_t P, const double* dirs3, const lrc_grid* grid,
                          double max_range, const lrc_frames* out, uint64_t capacity, uint64_t* out_total) {
    if (out_total) *out_total = 0;
    if (!grid) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_grid_compact: grid is NULL");
    const uint64_t N = (uint64_t)grid->lines * grid->width;
    int rc = check_grid("lrc_scan_grid_compact", grid, N);
    if (rc) return rc;
    return scan_compact_impl(s, poses16, P, dirs3, N, grid, max_range, out, capacity, out_total);
}

static int scan_compact_impl(lrc_scene* s, const double* poses16, uint64_t P, const double* dirs3, uint64_t N,
                             const lrc_grid* grid, double max_range, const lrc_frames* out, uint64_t capacity,
                             uint64_t* out_total, const double* d_dirs3) {
    if (out_total) *out_total = 0;
    if (!s || !out) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_poses_compact: NULL scene or output");
    const uint64_t n = P * N;
    if (!n) return LRC_OK;
    if (!poses16 || (!dirs3 && !d_dirs3) || !out->counts)
        return fail(LRC_ERR_INVALID_ARG, "lrc_scan_poses_compact: poses16, dirs3 or counts is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    SyncUnlessOk guard;
    DevBuf dp, dd;
    int rc;
    if ((rc = ensure_streams(s->ctx))) return rc;
    hipStream_t in = s->ctx->s_compute;       // inputs travel on the stream that consumes them
    if ((rc = dp.get(s->ctx, kPoolPoses, P * 128))) return rc;
    LRC_HIP(hipMemcpyAsync(dp.p, poses16, P * 128, hipMemcpyHostToDevice, in));
    if (d_dirs3) {
        dd.p = (void*)d_dirs3; dd.pooled = true;            // resident table (lrc_table): nothing to upload
    } else {
        if ((rc = dd.get(s->ctx, kPoolDirs, N * 24))) return rc;
        LRC_HIP(hipMemcpyAsync(dd.p, dirs3, N * 24, hipMemcpyHostToDevice, in));
    }
    FrameStage st;
    if ((rc = st.alloc(s->ctx, *out, P, n))) return rc;
    NoiseStage ns;
    if ((rc = ns.begin(s, n))) return rc;
    TraceParams p{};
    p.poses16 = (const double*)dp.p;
    p.dirs3 = (const double*)dd.p;
    p.rays_per_pose = N;
    p.total = n;
    p.has_center = 1;
    p.max_range = max_range;
    s->cur_grid = grid;
    rc = frames_finish(s, p, grid ? 3 : 1, st, P, N, out, capacity, out_total);
    s->cur_grid = nullptr;
    return guard.done(rc);
}

int lrc_scan_angles_dev(lrc_scene* s, const double* d_poses16, uint64_t P, const double* d_angles2,
                        const uint8_t* d_keep, uint64_t N, double max_range, const lrc_hits* d_out, void* stream) {
    if (!s || !d_out) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_angles_dev: NULL scene or output");
    if (P && N && (!d_poses16 || !d_angles2))
        return fail(LRC_ERR_INVALID_ARG, "lrc_scan_angles_dev: poses16 or angles2 is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    TraceParams p{};
    p.poses16 = d_poses16;
    p.angles2 = d_angles2;
    p.keep_mask = d_keep;
    p.rays_per_pose = N ? N : 1;
    p.total = P * N;
    p.has_center = 1;
    p.max_range = max_range;
    p.out = *d_out;
    return launch_trace(s, p, 2, (hipStream_t)stream);
}

int lrc_scan_angles_compact(lrc_scene* s, const double* poses16, uint64_t P, const double* angles2,
                            const uint8_t* keep, uint64_t N, double max_range, const lrc_frames* out,
                            uint64_t capacity, uint64_t* out_total) {
    if (out_total) *out_total = 0;
    if (!s || !out) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_angles_compact: NULL scene or output");
    const uint64_t n = P * N;
    if (!n) return LRC_OK;
    if (!poses16 || !angles2 || !out->counts)
        return fail(LRC_ERR_INVALID_ARG, "lrc_scan_angles_compact: poses16, angles2 or counts is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    SyncUnlessOk guard;
    DevBuf dp, da, dk;
    int rc;
    if ((rc = ensure_streams(s->ctx))) return rc;
    hipStream_t in = s->ctx->s_compute;
    if ((rc = dp.get(s->ctx, kPoolPoses, P * 128)) || (rc = da.get(s->ctx, kPoolAngles, n * 16))) return rc;
    LRC_HIP(hipMemcpyAsync(dp.p, poses16, P * 128, hipMemcpyHostToDevice, in));
    LRC_HIP(hipMemcpyAsync(da.p, angles2, n * 16, hipMemcpyHostToDevice, in));
    if (keep) {
        if ((rc = dk.get(s->ctx, kPoolKeep, n))) return rc;
        LRC_HIP(hipMemcpyAsync(dk.p, keep, n, hipMemcpyHostToDevice, in));
    }
    FrameStage st;
    if ((rc = st.alloc(s->ctx, *out, P, n))) return rc;
    NoiseStage ns;
    if ((rc = ns.begin(s, n))) return rc;
    TraceParams p{};
    p.poses16 = (const double*)dp.p;
    p.angles2 = (const double*)da.p;
    p.keep_mask = keep ? (const uint8_t*)dk.p : nullptr;
    p.rays_per_pose = N;
    p.total = n;
    p.has_center = 1;
    p.max_range = max_range;
    return guard.done(frames_finish(s, p, 2, st, P, N, out, capacity, out_total));
}

int lrc_scan_rays_compact(lrc_scene* s, const float* rays6, const uint8_t* keep, const double* centers3, uint64_t P,
                          uint64_t N, double max_range, const lrc_frames* out, uint64_t capacity, uint64_t* out_total) {
    if (out_total) *out_total = 0;
    if (!s || !out) return fail(LRC_ERR_INVALID_ARG, "lrc_scan_rays_compact: NULL scene or output");
    const uint64_t n = P * N;
    if (!n) return LRC_OK;
    if (!rays6 || !centers3 || !out->counts)
        return fail(LRC_ERR_INVALID_ARG, "lrc_scan_rays_compact: rays6, centers3 or counts is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    SyncUnlessOk guard;
    DevBuf dr, dc, dk;
    int rc;
    if ((rc = ensure_streams(s->ctx))) return rc;
    hipStream_t in = s->ctx->s_compute;
    if ((rc = dr.get(s->ctx, kPoolRays, n * 24)) || (rc = dc.get(s->ctx, kPoolCen, P * 24))) return rc;
    LRC_HIP(hipMemcpyAsync(dr.p, rays6, n * 24, hipMemcpyHostToDevice, in));
    LRC_HIP(hipMemcpyAsync(dc.p, centers3, P * 24, hipMemcpyHostToDevice, in));
    if (keep) {
        if ((rc = dk.get(s->ctx, kPoolKeep, n))) return rc;
        LRC_HIP(hipMemcpyAsync(dk.p, keep, n, hipMemcpyHostToDevice, in));
    }
    FrameStage st;
    if ((rc = st.alloc(s->ctx, *out, P, n))) return rc;
    NoiseStage ns;
    if ((rc = ns.begin(s, n))) return rc;
    TraceParams p{};
    p.rays6 = (const float*)dr.p;
    p.seg_centers3 = (const double*)dc.p;
    p.keep_mask = keep ? (const uint8_t*)dk.p : nullptr;
    p.rays_per_pose = N;
    p.total = n;
    p.has_center = 1;
    p.max_range = max_range;
    return guard.done(frames_finish(s, p, 0, st, P, N, out, capacity, out_total));
}

int lrc_debug_scan_stats(lrc_scene* s, const double* poses16, uint64_t P, const double* dirs3, uint64_t N,
                         double max_range, uint32_t* stats) {
    if (!s || !stats) return fail(LRC_ERR_INVALID_ARG, "lrc_debug_scan_stats: NULL scene or output");
    const uint64_t n = P * N;
    if (!n) return LRC_OK;
    if (!poses16 || !dirs3) return fail(LRC_ERR_INVALID_ARG, "lrc_debug_scan_stats: poses16 or dirs3 is NULL");
    LRC_HIP(hipSetDevice(s->ctx->device));
    DevBuf dp, dd, ds;
    int rc;
    if ((rc = dp.get(s->ctx, kPoolPoses, P * 128)) || (rc = dd.get(s->ctx, kPoolDirs, N * 24)) ||
        (rc = ds.get(s->ctx, kPoolStats, n * kStatsWords * 4)))
        return rc;
    LRC_HIP(hipMemcpy(dp.p, poses16, P * 128, hipMemcpyHostToDevice));
    LRC_HIP(hipMemcpy(dd.p, dirs3, N * 24, hipMemcpyHostToDevice));
    TraceParams p{};
    p.poses16 = (const double*)dp.p;
    p.dirs3 = (const double*)dd.p;
    p.rays_per_pose = N;
    p.total = n;
    p.has_center = 1;
    p.max_range = max_range;
    p.stats = (uint32_t*)ds.p;
    if ((rc = launch_trace(s, p, 1, nullptr, true))) return rc;
    LRC_HIP(hipDeviceSynchronize());
    LRC_HIP(hipMemcpy(stats, ds.p, n * kStatsWords * 4, hipMemcpyDeviceToHost));
    return LRC_OK;
}

}  // extern "C"

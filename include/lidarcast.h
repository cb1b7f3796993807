/*
 * lidarcast.h -- C ABI of liblidarcast, the MI355X (gfx950) LiDAR ray-cast scan engine.
 *
 * The reference has no FFI on this path: its engine boundary is the Python ABC
 * RaycastEngineBase (raycast_engine/raycast_engine.py:16-61) whose concrete classes call
 * open3d.t.geometry.RaycastingScene (raycast_engine/raycast_engine_cpu.py:46-51,
 * raycast_engine/raycast_engine_gpu_simple.py:41-46).  The entry points below are what a
 * ctypes binding of that boundary needs ("Suggested C exports"); each
 * one names the reference call it stands in for.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every function returns an int status: 0 (LRC_OK) or a negative LRC_ERR_* code;
 *     lrc_last_error() returns a thread-local, NUL-terminated description of the last failure.
 *   - no exception, abort() or longjmp crosses this boundary.
 *   - the caller allocates inputs and outputs; the library owns only the opaque handles.
 *   - "host" entry points take host pointers, copy, run the HIP kernels and copy back
 *     (synchronous on return).  "_dev" entry points take DEVICE pointers and a hipStream_t
 *     (passed as void*; NULL = the null stream) and only enqueue work.
 *   - handles are not thread-safe; use one context per thread.
 *   - finite-ray contract: a ray with a NaN or infinite origin or direction component is never cast; it is reported
 *     as a miss (t = +inf, prim = LRC_INVALID_PRIM, zeros elsewhere), as Embree reports a ray it cannot intersect.
 *     Mesh vertices must be finite and within 1e6 (lrc_scene_create rejects others).
 *
 * Hit definition (oracle/lrc_oracle.c restates it on the CPU)
 *   Two-sided closest hit, t in (0, +inf), t parametric along the GIVEN direction (the direction
 *   is not normalised, as in Open3D), miss = +inf / prim 0xFFFFFFFF, ties broken by the smaller
 *   triangle row index.  float32 throughout with a fixed fused-multiply-add expression tree.
 */
#ifndef LIDARCAST_H
#define LIDARCAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRC_OK                 0
#define LRC_ERR_INVALID_ARG  (-1)   /* NULL pointer, bad size, triangle index out of range      */
#define LRC_ERR_NO_DEVICE    (-2)   /* no HIP device / device ordinal out of range              */
#define LRC_ERR_HIP          (-3)   /* a HIP runtime call failed; text in lrc_last_error()      */
#define LRC_ERR_OOM          (-4)   /* host or device allocation failed                         */
#define LRC_ERR_INTERNAL     (-5)

#define LRC_INVALID_PRIM  0xFFFFFFFFu

typedef struct lrc_ctx   lrc_ctx;     /* one HIP device + scratch buffers                       */
typedef struct lrc_scene lrc_scene;   /* one triangle mesh + its BVH, resident in HBM           */

/* Per-ray hit attributes, structure-of-arrays.  Every pointer may be NULL (attribute skipped).
 * Arrays hold one entry per ray (cast) or P*N entries, pose-major (scan).
 * Replaces the result dict of RaycastingScene.cast_rays (raycast_engine_cpu.py:51-53:
 * t_hit, primitive_ids, primitive_normals) plus the numpy post-processing that follows it. */
typedef struct lrc_hits {
    float*    t;             /* (n)    parametric range; +inf = miss or removed by the range filter */
    uint32_t* prim;          /* (n)    triangle row index in the caller's array; LRC_INVALID_PRIM   */
    float*    normal3;       /* (n,3)  unit geometric normal normalize((v1-v0)x(v2-v0)); 0 on miss   */
    float*    point3;        /* (n,3)  o + (d/|d|)*t in float32, mul then add
                                       (raycast_engine_cpu.py:57-62); 0 on miss                      */
    uint16_t* sem;           /* (n)    semantic label of the hit triangle; 0 on miss                 */
    uint16_t* ins;           /* (n)    instance label of the hit triangle; 0 on miss                 */
    double*   incident_deg;  /* (n)    degrees(arccos(|((p-c)/|p-c|)_z|)) in float64
                                       (raycast_engine_cpu.py:100-107); 0 on miss                    */
    void*     t_label;       /* (n) x 8 B  DEVICE entry points only: packed {float t; uint32 sem|ins<<16}, the
                                       pair lrc_cloud_from_ranges_dev needs to rebuild a hit point          */
    uint32_t* tile_count;    /* (ceil(n/64)) DEVICE entry points only: number of kept entries in each
                                       aligned run of 64 outputs; lets lrc_compact_dev skip its counting
                                       pass (lrc_compact_io.tile_count).  Ignored by the host entry points */
    float*    intensity;     /* (n)    opt-in (the reference computes no intensity; its noise model expects one
                                       in [0,1], lidar/lidar_intrinsics.py:364-377): Lambertian return
                                       |cos(angle between the ray and the unit geometric normal)|, float32;
                                       0 on miss                                                        */
} lrc_hits;

typedef struct lrc_scene_info {
    uint64_t num_vertices;
    uint64_t num_triangles;
    uint64_t num_nodes;        /* inner nodes of the binary BVH (64 B each)                      */
    uint64_t num_leaves;
    uint64_t num_slots;        /* triangle records in leaf order (48 B each), == num_triangles   */
    uint32_t max_depth;        /* root = depth 0; bounded by LRC_MAX_BVH_DEPTH                   */
    uint32_t max_leaf_size;
    uint64_t device_bytes;     /* HBM held by this scene                                         */
    double   build_ms;         /* host BVH build time                                            */
    double   upload_ms;
    float    bounds_lo[3];
    float    bounds_hi[3];
    uint32_t quantised_nodes;  /* 1: the trace kernels walk the 32-byte quantised node images of this tree   */
    float    leaf_inflation;   /* mean half perimeter of a leaf box on the 15-bit grid / of its float32 box  */
    uint32_t device_build;     /* 1: the BVH was built on the GPU (build_ms = hierarchy + layout kernels,
                                  upload_ms = mesh transfer + validation); 0: host builder                     */
    uint32_t reserved_;
} lrc_scene_info;

#define LRC_MAX_BVH_DEPTH 32

/* Library version string, e.g. "lidarcast 0.1.0 (gfx950)". */
const char* lrc_version(void);

/* Thread-local text of the last error raised on this thread ("" if none). */
const char* lrc_last_error(void);

/* Number of visible HIP devices (0 when there is none); never fails. */
int lrc_device_count(void);

/* Bind a context to HIP device `device`.  Fails with LRC_ERR_NO_DEVICE on a box without a GPU so
 * that RaycastEngineGPU() raises and the caller's try/except (s3dis_simulator.py:66-74) can act. */
int lrc_ctx_create(int device, lrc_ctx** out_ctx);
int lrc_ctx_destroy(lrc_ctx* ctx);
int lrc_ctx_synchronize(lrc_ctx* ctx);

/* Launch chaining (opt-in; available where the device supports stream memory operations).
 * The poses of a trajectory are independent (s3dis_simulator.py:254-288 carries no state from one waypoint to the
 * next), so a caller may keep two scans in flight: lrc_scan_poses_dev / lrc_scan_angles_dev / lrc_cast*_dev on two
 * streams with two record sets.  A scan launch ends in a tail -- a few long-running waves on an emptying chip -- and what
 * two unordered launches make of it is luck: side by side they end together (nothing gained) or staggered (+12 %).
 * With chaining the library orders them: a scan enqueued on ANOTHER stream than the previous scan of this context is held
 * (hipStreamWaitValue64 on a signal word the previous launch's last workgroup writes when it starts) until that launch
 * has no workgroup left to hand out, and then fills the slots its tail leaves empty.  Consecutive scans on ONE stream
 * are untouched (stream order already serialises them).  The only semantic effect is an extra ordering edge from the
 * earlier-enqueued scan to the later one; results are unaffected.  Off by default: on MI355X the workgroup dispatcher
 * serves one launch of one-wave workgroups at a time anyway, so two bare trace launches on two
 * streams already overlap this way and the explicit order measured equal (tools/pipe_time.py, tools/two_in_flight.py);
 * it is kept for callers whose streams carry other kernels between the scans. */
int lrc_ctx_set_launch_chaining(lrc_ctx* ctx, int enabled);
int lrc_ctx_get_launch_chaining(const lrc_ctx* ctx, int* out_enabled, int* out_supported);

/* Build the scene once per mesh: float32 vertices (V,3) and uint32 triangle rows (T,3) in, binned-SAH
 * BVH (leaves <= 4 triangles, bounded depth), triangle records, id / label / plane tables and the
 * quantised node images out, all resident in HBM.  tri_sem / tri_ins are optional per-triangle labels.
 * The build runs ON THE GPU (csrc/lrc_bvh_device.hip; milliseconds for a 10^6-triangle mesh); the
 * host builder (csrc/bvh_build.cpp) produces the same tree and the same bytes and serves meshes of a
 * handful of triangles and LRC_DEVICE_BUILD=0.
 * Replaces RaycastingScene() + TriangleMesh.from_legacy + add_triangles, which the reference
 * repeats on every call (raycast_engine_cpu.py:46-47, raycast_engine.py:20-24).
 * T == 0 is allowed (every ray misses). */
int lrc_scene_create(lrc_ctx* ctx,
                     const float* verts3, uint64_t num_vertices,
                     const uint32_t* tris3, uint64_t num_triangles,
                     const uint16_t* tri_sem, const uint16_t* tri_ins,
                     lrc_scene** out_scene);
/* The same for a mesh that is already in HBM (DEVICE pointers; labels may be NULL): nothing crosses
 * PCIe.  Synchronous like lrc_scene_create; the input arrays may be released on return. */
int lrc_scene_create_dev(lrc_ctx* ctx,
                         const float* d_verts3, uint64_t num_vertices,
                         const uint32_t* d_tris3, uint64_t num_triangles,
                         const uint16_t* d_tri_sem, const uint16_t* d_tri_ins,
                         lrc_scene** out_scene);
int lrc_scene_destroy(lrc_scene* scene);
int lrc_scene_get_info(const lrc_scene* scene, lrc_scene_info* out_info);

/* Copy the BVH back to host arrays (tests check its invariants).  Any pointer may be NULL.
 *   nodes16 : num_nodes * 16 floats  (lo0 xyz, hi0 xyz, lo1 xyz, hi1 xyz, ref0, ref1, 0, 0; refs are
 *             int32 bit patterns: >=0 inner node index, <0 leaf: ~ref = first_slot*8 + count)
 *   slot_prim: num_slots uint32 (triangle row index stored in each leaf slot) */
int lrc_scene_export_bvh(const lrc_scene* scene, float* nodes16, uint32_t* slot_prim);

/* Copy one of the scene's device arrays back (tests compare the device builder's bytes with the host builder's).
 * dst may be NULL to ask for the size only (*out_bytes; 0 for an array this scene does not have). */
#define LRC_ARRAY_NODES       0   /* num_nodes x 64 B                                      */
#define LRC_ARRAY_TRIS        1   /* num_slots x 48 B: v0 v1 v2 Ng                         */
#define LRC_ARRAY_SLOT_PRIM   2   /* num_slots x u32                                       */
#define LRC_ARRAY_SLOT_LABEL  3   /* num_slots x u32: sem | ins << 16                      */
#define LRC_ARRAY_PRIM_PLANE  4   /* num_triangles x 32 B: (v0, label bits), (Ng, 0)       */
#define LRC_ARRAY_NODES_Q     5   /* num_nodes x 32 B quantised image                      */
#define LRC_ARRAY_NODES_N     6   /* num_nodes x 64 B normalised float32 image             */
int lrc_scene_export_array(const lrc_scene* scene, int which, void* dst, uint64_t dst_bytes, uint64_t* out_bytes);

/* ---- cast: explicit rays --------------------------------------------------------------------
 * rays6 is (N,6) float32 rows [ox,oy,oz,dx,dy,dz] (raycast_engine_cpu.py:24-38).
 * center3 (3 doubles, may be NULL) and max_range implement lidar_intersect_mesh's range filter
 * (raycast_engine_cpu.py:95-97): a hit is kept iff |float64(point) - center| < max_range (strict);
 * removed hits are reported exactly like misses.  With center3 == NULL nothing is filtered and
 * incident_deg is measured from the ray origin.
 * Replaces RaycastingScene.cast_rays + the numpy block raycast_engine_cpu.py:50-73. */
int lrc_cast(lrc_scene* scene, const float* rays6, uint64_t num_rays,
             const double* center3, double max_range, const lrc_hits* out);
int lrc_cast_dev(lrc_scene* scene, const float* d_rays6, uint64_t num_rays,
                 const double* center3 /* HOST pointer, 3 doubles or NULL */, double max_range,
                 const lrc_hits* d_out, void* stream);

/* ---- cast: explicit rays of several poses in one launch ------------------------------------------
 * rays6 holds the rays of S poses back to back (ragged: the dual-axis sensor drops ~2 % of its rays per
 * pose, lidar/indoor_lidar.py:292-294); seg_offsets (S+1 entries, first 0, last num_rays) delimits them
 * and centers3 (S,3) gives each pose's range-filter centre.  Per ray exactly lrc_cast with its pose's
 * centre.  Replaces the per-waypoint loop body (s3dis_simulator.py:254-264) for sensors whose rays are
 * generated on the host. */
int lrc_cast_segments(lrc_scene* scene, const float* rays6, uint64_t num_rays,
                      const uint64_t* seg_offsets, uint64_t num_segments, const double* centers3,
                      double max_range, const lrc_hits* out);
int lrc_cast_segments_dev(lrc_scene* scene, const float* d_rays6, uint64_t num_rays,
                          const uint64_t* d_seg_offsets, uint64_t num_segments,
                          const double* d_centers3, double max_range, const lrc_hits* d_out,
                          void* stream);

/* ---- scan: pose-batched, rays generated in the kernel -------------------------------------------
 * poses16: (P,16) float64 row-major 4x4 sensor poses (Waypoint.to_pose_matrix,
 *          trajectory/trajectory_generator.py:30-44).
 * dirs3  : (N,3) float64 sensor-frame unit directions, line-major/azimuth-minor
 *          (IndoorLidar._gen_lidar_rays_with_vertical_degrees, lidar/indoor_lidar.py:108-126).
 * Ray (p,i): origin = float32(pose[:3,3]); direction = float32(dirs3[i] @ R^T) evaluated in float64 as the
 * fused chain fma(c,R[j][2], fma(b,R[j][1], a*R[j][0])) -- what numpy's BLAS product gives bit for bit
 * (lidar/indoor_lidar.py:127-131); then exactly lrc_cast with
 * center = pose[:3,3] and max_range.  Output index = p*N + i.
 * Replaces the per-waypoint loop body s3dis_simulator.py:254-264. */
int lrc_scan_poses(lrc_scene* scene, const double* poses16, uint64_t num_poses,
                   const double* dirs3, uint64_t rays_per_pose, double max_range,
                   const lrc_hits* out);
int lrc_scan_poses_dev(lrc_scene* scene, const double* d_poses16, uint64_t num_poses,
                       const double* d_dirs3, uint64_t rays_per_pose, double max_range,
                       const lrc_hits* d_out, void* stream);

/* ---- compaction: fixed-stride records -> the reference's variable-length frames ----------------
 * Stable compaction, in (segment, ray) order, of the entries whose t is finite.  `t` holds
 * num_segments * seg_len entries.  counts[s] receives the number kept in segment s;
 * the kept entries of all segments are packed back to back, which is the order of
 * np.vstack over frames (containers/s3dis_sim_scene.py:326,362).  Optional gathers (NULL = skip):
 * point3 -> out_point3 (K,3), sem/ins -> out_sem/out_ins (K), incident -> out_incident (K),
 * out_index (K) = index of the kept entry inside its segment, out_xyzl (K,4) = point + packed labels.
 * Host variant returns the total K in *out_total. */
typedef struct lrc_compact_io {
    const float*    t;
    const float*    point3;
    const uint16_t* sem;
    const uint16_t* ins;
    const double*   incident_deg;
    const uint32_t* tile_count;    /* optional, device entry point only: lrc_hits.tile_count of the scan
                                      that produced t (used when seg_len % 64 == 0)                   */
    uint64_t*       counts;        /* (num_segments)                                  */
    float*          out_point3;
    uint16_t*       out_sem;
    uint16_t*       out_ins;
    double*         out_incident_deg;
    uint32_t*       out_index;
    float*          out_xyzl;      /* (K,4) packed rows x, y, z, label bits (sem | ins<<16): the 16-byte
                                      row the multi-GPU all-gather moves                              */
    float*          out_range_origin; /* (K) |point| from the WORLD origin in float32, formed as numpy's
                                      np.linalg.norm(points, axis=1) forms it: the quantity the reference takes its
                                      ScanQuality range statistics over (s3dis_simulator.py:283-284)  */
} lrc_compact_io;

int lrc_compact(lrc_ctx* ctx, uint64_t num_segments, uint64_t seg_len,
                const lrc_compact_io* io, uint64_t* out_total);
int lrc_compact_dev(lrc_ctx* ctx, uint64_t num_segments, uint64_t seg_len,
                    const lrc_compact_io* d_io, void* stream);

/* ---- the scan pipeline: consecutive pose batches of one scene, launches overlapped inside the library ---------------
 * The poses of a trajectory are independent (s3dis_simulator.py:254-288), and so are consecutive trajectories over one
 * mesh (the reference's batch driver, s3dis_simulator.py:594-726, runs them one after the other).  A caller that stays on
 * the device -- dataset generation, the multi-GPU step, bench.py -- submits batch after batch; each submit is one pose-batched
 * scan (lrc_scan_poses_dev: the complete 36-byte record per ray, into one of four record sets the pipeline owns) plus the
 * stable compaction (lrc_compact_dev) into the CALLER's rows / counts.  The pipeline arranges the launches so that the trace
 * of submit k+1 fills the wave slots the trace of submit k leaves empty while its last, long-running waves finish, and the
 * rows of submit k are scattered by the first workgroups of the trace launch of submit k+2 (DESIGN.md "the launch tail":
 * a 64-pose launch alone loses a sixth of its time to that tail).  Same arithmetic, same bytes as lrc_scan_poses_dev +
 * lrc_compact_dev called one after the other.
 *   lrc_pipe_submit   enqueues only.  Inputs and the output buffers are taken as of `stream`'s current position; nothing
 *                     is ordered after it on `stream` -- that is the point -- until
 *   lrc_pipe_wait     scatters the rows still in the pipeline and makes `stream` wait for every submit so far (event
 *                     waits, no host synchronisation).  Outputs of a submit are complete once `stream` has passed it.
 *   d_out             the out_* members and counts of lrc_compact_io (device pointers); the input members are ignored.
 *                     The buffers of a submit are written up to two submits later: rotate at least three output buffers
 *                     between lrc_pipe_wait calls.  rays_per_pose % 64 != 0 falls back to scan + compaction per stream.
 *   lrc_pipe_records  the fixed-stride records (lrc_hits, device pointers, tile_count included) of submit `ticket`;
 *                     valid until three further submits have been made (four sets rotate).
 *   lrc_pipe_trace_ms the time the trace launch of submit `ticket` spent between its stream reaching it and its last wave
 *                     (HIP events on the launch stream; inside the pipeline launches overlap, so this is longer than the
 *                     launch's share of the step).  Synchronises the host with that launch.
 * Destroy the pipeline before its scene. */
typedef struct lrc_pipe lrc_pipe;
int lrc_pipe_create(lrc_scene* scene, uint64_t max_poses, uint64_t rays_per_pose, lrc_pipe** out_pipe);
int lrc_pipe_destroy(lrc_pipe* pipe);
int lrc_pipe_submit(lrc_pipe* pipe, const double* d_poses16, uint64_t num_poses, const double* d_dirs3, double max_range,
                    const lrc_compact_io* d_out, void* stream, uint64_t* out_ticket);
int lrc_pipe_wait(lrc_pipe* pipe, void* stream);
int lrc_pipe_records(lrc_pipe* pipe, uint64_t ticket, lrc_hits* out_records);
int lrc_pipe_trace_ms(lrc_pipe* pipe, uint64_t ticket, float* out_ms);

/* The pipeline on N ranks (one process per GPU; the collective itself is the caller's: RCCL all-gather of the send slabs).
 *   lrc_pipe_submit_sharded  traces this rank's pose block like lrc_pipe_submit, with the triangle ids and per-wave keep counts
 *                            written straight into the caller's send slab (d_send_prim: P * rays_per_pose words,
 *                            d_send_tile_count: one word per 64 rays); `assemble` (nullable) describes an EARLIER scan whose
 *                            slabs of ALL ranks have been gathered: its assembly -- the other ranks' rows rebuilt from their
 *                            ids (lrc_cloud_from_prims_own_dev's arithmetic, bit-identical), this rank's own rows scattered
 *                            from the records of submit `own_ticket` -- rides in the leading workgroups of this trace launch,
 *                            i.e. it runs when the previous launch's tail begins instead of waiting, as a separate kernel of
 *                            4-wave workgroups does, until a trace launch has no workgroup left.  `stream` must already wait
 *                            for the collective that produced the gathered slabs.
 *   lrc_pipe_trace_done      `stream` waits for the trace of that submit (the send slab is complete: start the collective).
 *   lrc_pipe_scan_gathered   the scan over the gathered keep counts of ALL ranks, to be enqueued on the communication stream
 *                            right behind the collective (two small kernels; the launch that carries the assembly must wait for
 *                            them through `stream` of lrc_pipe_submit_sharded).
 *   lrc_pipe_assemble        the same assembly with the plain kernels on `stream` (the last scans of a run), after the scan.
 * Rows and counts of an assembled scan are complete once the stream has passed lrc_pipe_wait (or lrc_pipe_assemble). */
typedef struct lrc_gathered {
    const double*   d_all_poses16;     /* (num_poses_all,16) the poses of ALL ranks, slab after slab                    */
    uint64_t        num_poses_all;     /* = slabs * poses_per_slab (a rank with fewer poses pads its slab with invalid ids) */
    const uint32_t* d_all_prims;       /* entry 0 of slab 0 of the gathered triangle ids                                 */
    const uint32_t* d_all_tile_counts; /* entry 0 of slab 0 of the gathered per-wave keep counts                         */
    uint64_t        poses_per_slab;
    uint64_t        slab_stride_bytes; /* distance between slabs (ids and counts share it)                               */
    uint64_t        own_slab;          /* this rank's slab                                                               */
    uint64_t        own_ticket;        /* the submit that scanned it (its records must still be there: <= 3 submits ago) */
    uint64_t        scan_slot;         /* 0 / 1: which of the pipeline's two offset tables lrc_pipe_scan_gathered fills for
                                          this scan (alternate with the gather buffers)                                  */
    float*          d_out_xyzl;        /* (K,4) the assembled rows, np.vstack order                                      */
    uint64_t*       d_counts;          /* (num_poses_all) kept rays per pose, nullable                                   */
} lrc_gathered;
int lrc_pipe_submit_sharded(lrc_pipe* pipe, const double* d_poses16, uint64_t num_poses, const double* d_dirs3, double max_range,
                            uint32_t* d_send_prim, uint32_t* d_send_tile_count, const lrc_gathered* assemble, void* stream,
                            uint64_t* out_ticket);
int lrc_pipe_trace_done(lrc_pipe* pipe, uint64_t ticket, void* stream);
int lrc_pipe_scan_gathered(lrc_pipe* pipe, const double* d_dirs3, const lrc_gathered* gathered, void* stream);
int lrc_pipe_assemble(lrc_pipe* pipe, const double* d_dirs3, const lrc_gathered* gathered, void* stream);

/* ---- scan straight to the reference's variable-length frames ---------------------------------------
 * What S3DISSimulator.run_simulation needs from a whole trajectory (s3dis_simulator.py:254-288): per pose the kept
 * points and their attributes, in ray order.  One call = pose-batched scan (lrc_scan_poses_dev) + stable compaction
 * (lrc_compact_dev) in HBM + the per-pose counts + ONE transfer per requested array of exactly the K kept rows.
 * Frame p is rows [sum(counts[:p]), sum(counts[:p+1])) of every array: np.vstack order
 * (containers/s3dis_sim_scene.py:326).  All pointers are HOST pointers the caller allocated, `capacity` rows each
 * (num_poses * rays_per_pose always suffices; a smaller buffer fails with LRC_ERR_INVALID_ARG and *out_total = K, so
 * that the caller can retry).  counts is required, every other array may be NULL.  Buffers from lrc_host_alloc are
 * page-locked: the transfers then run as DMA at PCIe rate instead of through the runtime's staging copies. */
typedef struct lrc_frames {
    uint64_t* counts;        /* (num_poses) kept rays per pose                                       */
    float*    point3;        /* (K,3)                                                                */
    uint16_t* sem;           /* (K)                                                                  */
    uint16_t* ins;           /* (K)                                                                  */
    double*   incident_deg;  /* (K)                                                                  */
    uint32_t* index;         /* (K) index of the kept ray inside its pose (the surviving-ray list)   */
    float*    xyzl;          /* (K,4) x, y, z, label bits (sem | ins<<16)                            */
    float*    range_origin;  /* (K) see lrc_compact_io.out_range_origin                              */
    /* per-pose statistics, (num_poses) each, computed on the device with numpy's own arithmetic (pairwise summation in
     * 8192-element buffer chunks, every operation in the column's type: csrc/lrc_stats.h), so that they carry the bits
     * np.mean / np.std give on the same frame -- what the reference's ScanQuality records hold
     * (s3dis_simulator.py:276-286); 0 for a pose that kept nothing.  The column itself need not be requested. */
    float*    range_origin_mean;   /* np.mean(np.linalg.norm(points, axis=1)), float32               */
    float*    range_origin_std;    /* np.std(...)                                                    */
    double*   incident_mean;       /* np.mean(incident_angles), float64                              */
    double*   incident_std;        /* np.std(incident_angles)                                        */
} lrc_frames;
int lrc_scan_poses_compact(lrc_scene* scene, const double* poses16, uint64_t num_poses,
                           const double* dirs3, uint64_t rays_per_pose, double max_range,
                           const lrc_frames* out, uint64_t capacity, uint64_t* out_total);

/* ---- scan of a GRID sensor: one wavefront per packet of rays ------------------------------------------------
 * The multi-line sensor's rays form a grid: dirs3[j * width + i] = (cos a_j cos b_i, cos a_j sin b_i, sin a_j) with one
 * elevation a_j per scan line and b_i = az0 + i * az_step covering one turn (IndoorLidar with listed elevations,
 * lidar/indoor_lidar.py:94-131: az0 = pi, az_step = -2 pi / width).  Telling the library so lets it trace a whole packet
 * of rays (one pose x up to 8 lines x 64 azimuths) per wavefront -- the tree is walked once per packet for the
 * packet's frustum, and each surviving triangle runs the exact ray/triangle test only on the few rays whose direction
 * its bounding sphere can contain (csrc/lrc_sector.h) -- instead of once per ray.  Same hit definition, same result
 * BYTES as lrc_scan_poses_dev on the same table (tests assert it); the table itself still supplies the exact float64
 * directions.  Needs width % 64 == 0, width >= 256 and |az_step| * width = 2 pi; the caller vouches that dirs3 HAS
 * this structure (the Python engine derives az0 / az_step from the table and verifies every entry). */
typedef struct lrc_grid {
    uint32_t lines;      /* scan lines H                               */
    uint32_t width;      /* azimuths per line W                        */
    double   az0;        /* azimuth of column 0, radians               */
    double   az_step;    /* azimuth increment per column, radians      */
} lrc_grid;
int lrc_scan_grid_dev(lrc_scene* scene, const double* d_poses16, uint64_t num_poses, const double* d_dirs3,
                      const lrc_grid* grid, double max_range, const lrc_hits* d_out, void* stream);
int lrc_scan_grid_compact(lrc_scene* scene, const double* poses16, uint64_t num_poses, const double* dirs3,
                          const lrc_grid* grid, double max_range, const lrc_frames* out, uint64_t capacity,
                          uint64_t* out_total);

/* ---- a sensor's direction table resident in HBM -------------------------------------------------------------
 * A caller that scans pose after pose with the same sensor (the reference's per-waypoint loop,
 * s3dis_simulator.py:254-264) would upload the same (N,3) float64 table with every call; a table handle uploads it
 * once.  lrc_scan_table_compact is lrc_scan_poses_compact (grid == NULL) or lrc_scan_grid_compact (grid != NULL) on
 * that resident table. */
typedef struct lrc_table lrc_table;
int lrc_table_create(lrc_ctx* ctx, const double* dirs3, uint64_t rays_per_pose, lrc_table** out_table);
int lrc_table_destroy(lrc_table* table);
int lrc_scan_table_compact(lrc_scene* scene, const double* poses16, uint64_t num_poses, const lrc_table* table,
                           const lrc_grid* grid /* nullable */, double max_range, const lrc_frames* out,
                           uint64_t capacity, uint64_t* out_total);

/* Page-locked host memory for the frame buffers above (hipHostMalloc / hipHostFree).  The caller owns it. */
int lrc_host_alloc(lrc_ctx* ctx, uint64_t bytes, void** out_ptr);
int lrc_host_free(lrc_ctx* ctx, void* ptr);

/* ---- dual-axis sensor: rays generated in the kernel from host-drawn scan angles (opt-in) ---------------
 * The reference's DualAxisLidar draws, per ray, a noisy azimuth phi and elevation theta from the global numpy stream
 * and drops ~2 % of the rays with one more uniform draw (lidar/indoor_lidar.py:262-294); that stream is what "seeded
 * identically" means, so the draws stay on the host.  What moves to the device is everything after them: sin/cos,
 * the rotation into the world frame (un-fused, as numpy evaluates it, :283-287) and the float32 narrowing.
 *   angles2 : (num_poses * rays_per_pose, 2) float64 (phi, theta), pose-major
 *   keep    : nullable (num_poses * rays_per_pose) bytes, 0 = ray dropped by the sensor: never cast, reported as a
 *             miss, so the compacted frames equal those of casting only the kept rays
 * Output index = p * rays_per_pose + i; range filter centre = pose[:3,3].  NOT bit-guaranteed against the host
 * generator: the device's double-precision sin/cos need not round like the host's libm (the tests count the rays
 * whose float32 direction differs).  The default path (host generator + lrc_cast_segments) stays bit-exact. */
int lrc_scan_angles_dev(lrc_scene* scene, const double* d_poses16, uint64_t num_poses, const double* d_angles2,
                        const uint8_t* d_keep, uint64_t rays_per_pose, double max_range, const lrc_hits* d_out,
                        void* stream);
int lrc_scan_angles_compact(lrc_scene* scene, const double* poses16, uint64_t num_poses, const double* angles2,
                            const uint8_t* keep, uint64_t rays_per_pose, double max_range, const lrc_frames* out,
                            uint64_t capacity, uint64_t* out_total);

/* ---- explicit rays of several poses straight to frames --------------------------------------------------------
 * The frame-producing form of lrc_cast_segments for sensors whose rays come from the host generator (the dual-axis
 * sensor's bit-exact default path): every pose contributes rays_per_pose rays at a fixed stride,
 *   rays6    : (num_poses * rays_per_pose, 6) float32, pose-major
 *   keep     : nullable (num_poses * rays_per_pose) bytes, 0 = ray dropped by the sensor (lidar/indoor_lidar.py:292-294):
 *              never cast, reported as a miss -- the frames equal those of casting only the kept rays
 *   centers3 : (num_poses, 3) float64 range-filter centres (pose[:3,3])
 * and the result is compacted in HBM exactly like lrc_scan_poses_compact's (same lrc_frames, same ordering, per-pose
 * statistics included).  `index` is then the ray's index among ALL rays_per_pose rays of its pose. */
int lrc_scan_rays_compact(lrc_scene* scene, const float* rays6, const uint8_t* keep, const double* centers3,
                          uint64_t num_poses, uint64_t rays_per_pose, double max_range, const lrc_frames* out,
                          uint64_t capacity, uint64_t* out_total);

/* ---- numpy's legacy seeded stream (row a7: the BLK2GO generator draws its noise from np.random) --------------------
 * The reference draws, per pose, two normals per ray and then one uniform per ray from the GLOBAL numpy stream
 * (lidar/indoor_lidar.py:257-296).  lrc_rng_scan_draws produces exactly the doubles RandomState.normal(loc, scale,
 * normals_per_pose) followed by RandomState.random_sample(uniforms_per_pose) would return, pose after pose, for
 * num_poses poses, and leaves *state where numpy's generator would stand (np.random.get_state() / set_state()
 * tuples map to this struct field by field): MT19937 words, random_sample doubles, polar-method normals with the
 * cached second value.  Host code, multi-threaded (threads <= 0: the cores of the host, at most 16); no GPU involved. */
typedef struct lrc_mt19937_state {
    uint32_t key[624];
    int32_t  pos;          /* 0..624, next word of key[] (624: the block is used up)              */
    int32_t  has_gauss;
    double   gauss;        /* the cached normal when has_gauss                                   */
} lrc_mt19937_state;
int lrc_rng_scan_draws(lrc_mt19937_state* state, uint64_t num_poses, uint64_t normals_per_pose,
                       uint64_t uniforms_per_pose, double loc, double scale, double* out_normals,
                       double* out_uniforms, int threads);

/* The rays of one dual-axis pose from the sines and cosines of its scan angles (numpy's own, which only numpy reproduces):
 * d = (cos(theta) cos(phi), cos(theta) sin(phi), sin(theta)) rotated as the reference rotates it, ray by ray and un-fused,
 * (d0 R[j][0] + d1 R[j][1]) + d2 R[j][2], narrowed to float32 beside the pose's origin (lidar/indoor_lidar.py:274-291).
 * pose16: row-major 4x4 float64; out_rays6: (n, 6) float32.  Host code: the dozen numpy passes this replaces were what
 * bounded the BLK2GO trajectory once the draws were native. */
int lrc_rays_from_trig(const double* cos_theta, const double* sin_theta, const double* cos_phi, const double* sin_phi,
                       uint64_t n, const double* pose16, float* out_rays6);

/* ---- diagnostics ---------------------------------------------------------------------------------------
 * Per-ray traversal counters of a pose-batched scan from an instrumented build of the trace kernel, host arrays.
 * stats: (num_poses * rays_per_pose, LRC_STATS_WORDS) uint32: [0] inner-node steps, [1] triangle tests, [2] node steps
 * taken wave-uniformly (scalar fetch), [3] node steps with no child hit, [4] triangles that pass every
 * Moeller-Trumbore condition and are rejected only by the hit definition's box clause -- the one
 * clause Embree does not have; the tests assert a total of 0 on the BASELINE configurations. */
#define LRC_STATS_WORDS 5
int lrc_debug_scan_stats(lrc_scene* scene, const double* poses16, uint64_t num_poses, const double* dirs3,
                         uint64_t rays_per_pose, double max_range, uint32_t* stats);

/* ---- opt-in sensor-realism options --------------------------------
 * The reference DECLARES these sensor parameters but never applies them (SURVEY.md F6, F7:
 * Indoor8LineLidarIntrinsics.add_noise has no caller, lidar/lidar_intrinsics.py:364-389; min_range is never
 * read by the engines; the "incident angle" ignores the surface, raycast_engine_cpu.py:100-107).  Everything
 * is OFF by default, which is the reference's behaviour and what the parity tests pin.  Options are sticky
 * on the scene handle until changed; pass NULL to reset.
 *   min_range      > 0: a hit is kept only if its float64 distance from the centre is >= min_range
 *   range_noise    additive noise in metres, one float per ray of the NEXT calls (drawn by the caller, e.g.
 *                  from a seeded numpy stream, so that seeded runs are reproducible bit for bit):
 *                  t' = t + noise (float32); t' <= 0 drops the return; point, range filter and incident
 *                  angle use t'.  Host entry points take a host array, _dev entry points a device array;
 *                  range_noise_len must equal the number of rays of the call.
 *   incident_mode  0: reference (angle between centre->point and the vertical axis)
 *                  1: angle between the ray and the hit triangle's normal, acos(|d^.n|) in degrees */
typedef struct lrc_scan_options {
    double       min_range;
    const float* range_noise;
    uint64_t     range_noise_len;
    int          incident_mode;
} lrc_scan_options;
int lrc_scene_set_options(lrc_scene* scene, const lrc_scan_options* opts);

/* ---- scene cloud from per-ray (t, label) pairs of a pose-batched scan ------------------------------
 * A pose-batched scan is a pure function of (poses16, dirs3): whoever holds those can rebuild the hit point of
 * ray (p,i) from its t alone, with the same float32 arithmetic as the scan itself (bit-identical rows).  The
 * multi-GPU assembly therefore all-gathers 8-byte (t, label) pairs (lrc_hits.t_label) instead of 16-byte rows
 * and every rank runs this function over the gathered pairs of all poses:
 *   d_t_label : (P*N) x {float t; uint32 label}, +inf = no return
 *   d_out_xyzl: (K,4) rows x, y, z, label bits, stable pose-major order (np.vstack of the frames,
 *               containers/s3dis_sim_scene.py:326); d_counts (P), nullable: kept rays per pose. */
int lrc_cloud_from_ranges_dev(lrc_ctx* ctx, const double* d_poses16, uint64_t num_poses,
                              const double* d_dirs3, uint64_t rays_per_pose, const void* d_t_label,
                              float* d_out_xyzl, uint64_t* d_counts, void* stream);

/* The same rebuild from 4 bytes per ray: the hit triangle's row (lrc_hits.prim, Open3D's primitive_ids of
 * raycast_engine_cpu.py:51; LRC_INVALID_PRIM = no return).  A closest hit is a pure function of (pose, direction,
 * triangle), so a rank that holds the scene replica recomputes t with the scan's own ray/triangle test, bit for
 * bit, and the point and the triangle's labels from it.  This is what the multi-GPU all-gather moves (DESIGN.md
 * section 6): xGMI links, not the kernels, bound the multi-GPU job.  Not available while a range_noise option is
 * set (the noise is not a function of the triangle): use lrc_cloud_from_ranges_dev then.
 *   d_prim       : entry (pose p, ray i) at word (p / poses_per_slab) * slab_stride_bytes/4
 *                  + (p % poses_per_slab) * rays_per_pose + i -- the gathered send slabs of several ranks in place;
 *                  poses_per_slab = 0 means one contiguous (num_poses * rays_per_pose) array
 *   d_tile_count : nullable; the senders' lrc_hits.tile_count (kept rays per aligned run of 64), laid out in the
 *                  same slabs: tile k of slab r at d_tile_count[r * slab_stride_bytes/4 + k].  Saves the counting
 *                  pass; needs rays_per_pose % 64 == 0
 *   ranks that own fewer poses than poses_per_slab pad their slab with LRC_INVALID_PRIM entries (zero counts); the
 *   pose table then has num_poses = slabs * poses_per_slab rows, padded ones arbitrary.
 * Scratch (tile offsets, transposed direction table) belongs to the scene's context: lrc_cloud_from_prims_dev and
 * lrc_cloud_from_ranges_dev share one set, lrc_compact_dev has its own, so a rebuild may run on one stream beside a
 * compaction on another, but two rebuilds (or two compactions) of one context must not overlap in time. */
int lrc_cloud_from_prims_dev(lrc_scene* scene, const double* d_poses16, uint64_t num_poses,
                             const double* d_dirs3, uint64_t rays_per_pose, const uint32_t* d_prim,
                             const uint32_t* d_tile_count, uint64_t poses_per_slab, uint64_t slab_stride_bytes,
                             float* d_out_xyzl, uint64_t* d_counts, void* stream);

/* The same when the caller's own slab of poses does not need rebuilding: its rows come from the local records the trace
 * kernel wrote (own->t, own->point3, own->sem, own->ins: fixed-stride arrays of poses_per_slab x rays_per_pose entries),
 * scattered to their place in the assembled cloud; only the other slabs are rebuilt from ids.  Same output bytes.
 * Needs the senders' per-wave keep counts (d_tile_count, rays_per_pose % 64 == 0). */
int lrc_cloud_from_prims_own_dev(lrc_scene* scene, const double* d_poses16, uint64_t num_poses, const double* d_dirs3,
                                 uint64_t rays_per_pose, const uint32_t* d_prim, const uint32_t* d_tile_count,
                                 uint64_t poses_per_slab, uint64_t slab_stride_bytes, uint64_t own_slab,
                                 const lrc_compact_io* own, float* d_out_xyzl, uint64_t* d_counts, void* stream);

/* Per-pose mean / std of |p| (float32, from the WORLD origin, numpy's arithmetic: lrc_stats.h) over the assembled
 * (x, y, z, label) rows of a scan: the ScanQuality range statistics (s3dis_simulator.py:283-286) of a cloud that was
 * assembled on the device.  d_range: scratch of max_rows floats (receives |p| per row). */
int lrc_cloud_range_stats_dev(lrc_ctx* ctx, const float* d_xyzl, const uint64_t* d_counts, uint64_t num_poses,
                              uint64_t max_rows, float* d_range, float* d_mean, float* d_std, void* stream);

/* ---- nearest annotated point --------------------------------------
 * Exact 1-nearest-neighbour lookup of float32 query points in a float64 annotated cloud, float64 distances,
 * ties to the smaller row.  Replaces sklearn NearestNeighbors(n_neighbors=1, algorithm='ball_tree')
 * .fit(annotated).kneighbors(points) at containers/s3dis_sim_scene.py:416-418, which the reference runs on
 * every frame's hit points at export time to attach colour / semantic / instance labels; also used once per
 * mesh to bake per-triangle labels (triangle centroid -> nearest annotated point) that the trace kernel then
 * writes back per ray.  cell_size <= 0 picks a grid spacing from the point density. */
typedef struct lrc_nn lrc_nn;
int lrc_nn_create(lrc_ctx* ctx, const double* points3, uint64_t num_points, double cell_size, lrc_nn** out_nn);
int lrc_nn_destroy(lrc_nn* nn);
int lrc_nn_query(lrc_nn* nn, const float* query3, uint64_t num_queries, uint32_t* out_index,
                 double* out_dist /* nullable */);
int lrc_nn_query_dev(lrc_nn* nn, const float* d_query3, uint64_t num_queries, uint32_t* d_out_index,
                     double* d_out_dist /* nullable */, void* stream);

/* ---- validation metrics ---------------------------------------------
 * The O(n*m) parts of the reference's sampled cloud metrics (evaluate_single_scene.py:55-111), host arrays:
 *   lrc_min_distances : out_min[i] = min_j |a_i - b_j| in float32 (sum of squares, one sqrt): the directed
 *                       term of compute_chamfer_distance (:81-96) and compute_hausdorff_distance (:98-111)
 *   lrc_rbf_kernel_sum: sum_ij exp(-gamma * max(|a_i|^2 + |b_j|^2 - 2 a_i.b_j, 0)), float64: one of the
 *                       three kernel sums of compute_mmd_sampled (:55-79) */
int lrc_min_distances(lrc_ctx* ctx, const float* a3, uint64_t n, const float* b3, uint64_t m, float* out_min);
int lrc_rbf_kernel_sum(lrc_ctx* ctx, const float* a3, uint64_t n, const float* b3, uint64_t m, double gamma,
                       double* out_sum);

/* ---- robot-cube occupancy for the trajectory planner ----------------
 * out_flags[q] = 1 iff some mesh vertex lies inside the axis-aligned cube [p_q - half, p_q + half] (float64,
 * inclusive), for all Q positions at once.  Replaces AutoTrajectoryGenerator._is_point_inside_mesh
 * (trajectory/auto_trajectory_generator.py:219-238), which the reference evaluates position by position over
 * every vertex for the free-space grid (:129-139) and for every waypoint of every candidate (:345-356). */
typedef struct lrc_occ lrc_occ;
int lrc_occ_create(lrc_ctx* ctx, const double* verts3, uint64_t num_vertices, lrc_occ** out_occ);
int lrc_occ_destroy(lrc_occ* occ);
int lrc_occ_query(lrc_occ* occ, const double* points3, uint64_t num_points, double half, uint8_t* out_flags);

/* Resident waves per CU the runtime grants the pose-batched trace kernel on this scene (its LDS stack is sized by
 * the tree depth), its VGPR count and LDS bytes per wave: the occupancy figure bench.py reports. */
int lrc_scene_get_occupancy(const lrc_scene* scene, int* waves_per_cu, int* vgprs, int* lds_bytes);

/* Number of traversal-kernel launches and rays issued on this scene so far (bench bookkeeping). */
int lrc_scene_get_counters(const lrc_scene* scene, uint64_t* launches, uint64_t* rays);

#ifdef __cplusplus
}
#endif
#endif /* LIDARCAST_H */

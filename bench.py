#!/usr/bin/env python3
"""bench.py -- rays/s of the LiDAR ray-cast scan on MI355X (config C3 of BASELINE.md).

One "step" = one pass of the hot path over the whole trajectory batch, inputs resident in HBM:
  trace kernel (rays generated in-kernel from 64 poses x the 32x2048 direction table, BVH
  traversal, hit write-back of t/prim/normal/point/sem/ins, range filter)  ->  stable compaction
  into the scene cloud (np.vstack order); for N > 1 instead: one RCCL all-gather of the hit triangle ids and the
  rebuild of the whole scene cloud on every GPU.
At N = 1 the steps go through the library's scan pipeline (lrc_pipe_*: consecutive batches, the trace launch of step k+1
fills the wave slots the launch of step k leaves empty in its tail, the rows of step k are scattered by the leading
workgroups of the launch of step k+2; same bytes as the two calls one after the other, checked in the run);
--serial times the two calls on one stream instead (the round-3 step; what the rocprofv3 / PMC passes profile).
Weak scaling: every rank scans its own 64 poses of a 64*N-pose trajectory over a replica of the scene.

Prints ONE JSON line (rank 0).  `roofline` prices the trace kernel against the roof that binds it -- vector-ALU issue
under divergence (lane-operations that did work / peak lane-operations), from the committed counter profile of this
binary and the kernel time measured live -- and reports the second, equally loaded pipe (the CU's vector-memory return
path) and the HBM-side traffic beside it; SURVEY.md section 8(d)'s per-ray byte model is kept as a labelled side field
only (it is not a lower bound on traffic).  `cpu_baseline` times the CPU oracle (a restatement: Open3D/Embree, the
reference's CPU path, is not installed) on a bounded sample of the same workload.  `config.scene_create_ms` is what
the scene costs before the first ray: mesh in host memory -> BVH resident in HBM, built on the GPU.
"""
import argparse
import json
import math
import os
import sys
import time

# the pool's host driver supports dmabuf IPC only: without this RCCL's buffer registration across ranks fails
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "indoor-point-cloud-datasets-controllable-generation-method-for-mobile-"
                         "robots-3d-scene-perception_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Vector-ALU lane-operation peak: 256 CUs x 4 SIMD-32 x 32 lanes per clock x 2.4 GHz (MI355X_MICROARCH.md: a wave64
# VALU instruction issues over 2 cycles on a SIMD-32).  One lane-operation = one lane of one VALU instruction.
VALU_PEAK_GLANEOPS = 256 * 4 * 32 * 2.4          # = 78 643.2 G lane-op/s
N_SIMDS, CLOCK_GHZ = 1024, 2.4
MIN_TIMED_SECONDS = 3.0      # the timed block of --steps steps is repeated until this much wall time has passed (the driver's
                             # GPU-busy sampler must be able to see the device busy: round 3's 0.25 s fell between its samples)
SCENE = "synth_A6_office2"
POSES_PER_GPU = 64


def bytes_per_ray(T, in_kernel_raygen=True):
    """Algorithmic bytes per ray, SURVEY.md section 8(d): ray in (24, dropped when rays are generated
    in-kernel) + hit record out (36) + one root-to-leaf descent of a binary BVH with <= 4 triangles per
    leaf (64 B per level) + one 4-triangle leaf (144)."""
    levels = math.ceil(math.log2(max(T, 8) / 4.0))
    return (0 if in_kernel_raygen else 24) + 36 + 64 * levels + 144


def pmc_profile(kernel_prefix, rays_per_launch, scene):
    """Counters of the trace kernel from the committed rocprofv3 PMC passes (profiles/pmc_latest.json, written by
    tools/pmc.sh on this same bench command).  The file carries the SHA-256 of the sources the profiled binary was
    built from (__graft_entry__.source_fingerprint); it is used only when that equals the fingerprint of the tree
    this process runs from, names this kernel and this workload -- otherwise None, and the roofline says so.  The
    counts of a launch (instructions, active lanes, bytes) are properties of binary + workload, not of the box; the
    TIME they are divided by is measured live in this run."""
    import __graft_entry__ as entry
    fp = entry.source_fingerprint()
    # one file per scene (tools/pmc.sh with PMC_SCENE), pmc_latest.json = the headline scene
    for name in (f"pmc_{scene}.json", "pmc_latest.json"):
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                t = json.load(f)
            if (t.get("source_sha256") != fp or t.get("rays_per_launch") != rays_per_launch
                    or t.get("scene") != scene or not t["kernel"].startswith(kernel_prefix)):
                continue
            c = t["counters"]
            for k in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "FETCH_SIZE", "WRITE_SIZE"):
                float(c[k])
            return c
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def c3_sensor():
    import dataclasses
    from lidar import Indoor8LineLidarIntrinsics
    return dataclasses.replace(Indoor8LineLidarIntrinsics.create_dense_32line(), horizontal_res=2048)


def c3_poses(rank, world):
    """64 poses per rank on the straight line x = 1..4 m, y = 2, z = 1, yaw 0 (what an auto trajectory
    looks like in the reference: pure translations at fixed height).  The 64*world poses of the whole
    job are evenly spaced on that line; rank r owns the r-th contiguous block."""
    from trajectory import line_trajectory, poses_from_waypoints
    wps = line_trajectory((1.0, 2.0, 1.0), (4.0, 2.0, 1.0), POSES_PER_GPU * world)
    return poses_from_waypoints(wps[rank * POSES_PER_GPU:(rank + 1) * POSES_PER_GPU])


def host_threads():
    """Threads for the CPU leg: the cores this process may run on, capped at 32 (a one-GPU box share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def cpu_baseline(mesh, sensor, poses, budget_s=18.0):
    """Reference-faithful CPU figure: per pose, host ray generation + BVH rebuild (the reference rebuilds
    the Embree scene on every call, raycast_engine_cpu.py:46-47) + cast + numpy post-processing, for as
    many poses of the workload as fit the time budget.  Also the build-once variant."""
    from lidar import create_lidar
    from oracle import np_oracle
    from oracle.c_oracle import OracleMesh
    threads = host_threads()
    n_per = sensor.vertical_res * sensor.horizontal_res
    t_start = time.perf_counter()
    done, t_faithful = 0, 0.0
    while done < len(poses) and time.perf_counter() - t_start < budget_s * 0.65:
        t0 = time.perf_counter()
        om = OracleMesh(mesh.vertices, mesh.triangles).build()       # per-pose scene rebuild
        np_oracle.lidar_intersect_mesh(om, create_lidar(sensor, poses[done]), threads=threads)
        om.free()
        t_faithful += time.perf_counter() - t0
        done += 1
    om = OracleMesh(mesh.vertices, mesh.triangles).build()
    t0 = time.perf_counter()
    done_once = 0
    while done_once < len(poses) and time.perf_counter() - t0 < budget_s * 0.3:
        np_oracle.lidar_intersect_mesh(om, create_lidar(sensor, poses[done_once]), threads=threads)
        done_once += 1
    t_once = time.perf_counter() - t0
    ref = open3d_leg(mesh, sensor, poses, budget_s * 0.5)
    if ref is not None:
        # Open3D is installed on this host: the reference's own CPU path (Embree through RaycastingScene) is the baseline,
        # the port's figures ride along
        ref.update({"port_value": done * n_per / t_faithful, "port_build_once_value": done_once * n_per / t_once,
                    "port_cores": threads})
        return ref
    return {
        "open3d": "not installed on this host (import open3d: ModuleNotFoundError) -- the reference's Embree path cannot be "
                  "timed; the figures below are the C restatement",
        "value": done * n_per / t_faithful, "unit": "rays/s", "cores": threads, "kind": "port",
        "sample": f"{done} of {len(poses)} poses x {n_per} rays of the C3 workload ({t_faithful:.1f} s of CPU work), "
                  f"scene (BVH) rebuilt per pose as the reference does; oracle/lrc_oracle.c (C, pthreads) + "
                  f"numpy ray generation and post-processing",
        "build_once_value": done_once * n_per / t_once,
        "build_once_sample": f"{done_once} poses in {t_once:.1f} s, BVH built once",
    }


def open3d_leg(mesh, sensor, poses, budget_s):
    """The reference's CPU path itself, when Open3D is installed on the measuring host (SURVEY
    section 8(d)): the sequence raycast_engine_cpu.py:46-73,95-107 runs per waypoint -- RaycastingScene() + from_legacy +
    add_triangles (the Embree build, repeated per pose as the reference repeats it), cast_rays, the numpy post-processing --
    written here against Open3D's public API (the reference's files do not travel to the GPU box), with this package's ray
    generator.  Returns None when Open3D is absent (it is on the pool's boxes: probed in round 4)."""
    try:
        import open3d as o3d
    except Exception:                                                    # noqa: BLE001 - absent or broken: no reference leg
        return None
    from lidar import create_lidar
    legacy = o3d.geometry.TriangleMesh(o3d.utility.Vector3dVector(np.asarray(mesh.vertices, dtype=np.float64)),
                                       o3d.utility.Vector3iVector(np.asarray(mesh.triangles, dtype=np.int32)))
    n_per = sensor.vertical_res * sensor.horizontal_res

    def one_pose(pose_m, scene_=None):
        lidar = create_lidar(sensor, pose_m)
        rays = lidar.get_rays()
        sc = scene_
        if sc is None:                                                   # raycast_engine_cpu.py:46-47
            sc = o3d.t.geometry.RaycastingScene()
            sc.add_triangles(o3d.t.geometry.TriangleMesh.from_legacy(legacy))
        ans = sc.cast_rays(o3d.core.Tensor(rays.astype(np.float32)))     # :50-51
        t = ans["t_hit"].numpy()
        hit = t != np.inf                                                # :54-73
        o, d = rays[hit, :3], rays[hit, 3:]
        pts = o + d / np.linalg.norm(d, axis=1, keepdims=True) * t[hit][:, None]
        c = lidar.pose[:3, 3]                                            # :95-107
        dist_ = np.linalg.norm(pts - c, axis=1)
        keep = dist_ < lidar.intrinsics.max_range
        pts = pts[keep]
        v = (pts - c) / np.linalg.norm(pts - c, axis=1, keepdims=True)
        return pts, np.degrees(np.arccos(np.abs(v[:, 2])))

    t_start, done, t_faithful = time.perf_counter(), 0, 0.0
    while done < len(poses) and time.perf_counter() - t_start < budget_s * 0.65:
        t0 = time.perf_counter()
        one_pose(poses[done])
        t_faithful += time.perf_counter() - t0
        done += 1
    sc = o3d.t.geometry.RaycastingScene()
    sc.add_triangles(o3d.t.geometry.TriangleMesh.from_legacy(legacy))
    t0, done_once = time.perf_counter(), 0
    while done_once < len(poses) and time.perf_counter() - t0 < budget_s * 0.3:
        one_pose(poses[done_once], sc)
        done_once += 1
    t_once = time.perf_counter() - t0
    return {
        "value": done * n_per / t_faithful, "unit": "rays/s", "cores": host_threads(), "kind": "reference",
        "open3d": o3d.__version__,
        "sample": f"{done} of {len(poses)} poses x {n_per} rays of the C3 workload ({t_faithful:.1f} s of CPU work): Open3D "
                  f"{o3d.__version__} RaycastingScene rebuilt per pose + cast_rays + the numpy post-processing of "
                  f"raycast_engine_cpu.py:54-107, all cores (Embree / TBB default)",
        "build_once_value": done_once * n_per / t_once,
        "build_once_sample": f"{done_once} poses in {t_once:.1f} s, RaycastingScene built once",
    }


def caller_path(scene, sensor, poses, dirs, mesh, reps=7):
    """What a HOST caller of the plugin surface gets on the same workload (never the bench `value`):
    caller_path_rays_per_s   RaycastEngineGPU.scan_frames = lrc_scan_poses_compact: poses + direction table in host
                             memory -> scan + compaction in HBM -> the kept rows (point, labels) of all 64 poses in
                             page-locked host memory, frames as views; PCIe both ways included
    run_simulation_rays_per_s  S3DISSimulator.run_simulation on top of it: 64 S3DISSimFrame objects with the
                             reference's per-frame ScanQuality statistics computed by numpy on the host."""
    import time as _t
    import lidarcast
    n = len(poses) * len(dirs)
    table = lidarcast.DirectionTable(scene.ctx, dirs)       # the engine keeps the sensor's table resident like this
    ts = []
    for _ in range(reps + 2):
        t0 = _t.perf_counter()
        fr = scene.scan_poses_compact(poses, table, sensor.max_range, want=("point3", "sem", "ins"))
        ts.append(_t.perf_counter() - t0)
        del fr
    out = {"caller_path_rays_per_s": n / float(np.median(ts[2:])), "caller_path_ms": float(np.median(ts[2:])) * 1e3}
    ts = []
    for _ in range(reps + 2):                                   # the same call for a caller who wants the points only
        t0 = _t.perf_counter()
        fr = scene.scan_poses_compact(poses, table, sensor.max_range, want=("point3",))
        ts.append(_t.perf_counter() - t0)
        del fr
    out["caller_path_points_only_ms"] = float(np.median(ts[2:])) * 1e3
    try:
        from raycast_engine import RaycastEngineGPU
        from s3dis_simulator import S3DISSimulator
        from trajectory import Waypoint
        sim = S3DISSimulator({"raycast_engine": {"use_gpu": True}})
        sim.lidar_config = sensor
        sim.load_scene(mesh, "bench")
        wps = [Waypoint(m[0, 3], m[1, 3], m[2, 3], yaw=0.0, timestamp=float(i)) for i, m in enumerate(poses)]
        ts = []
        for _ in range(9):
            t0 = _t.perf_counter()
            sc = sim.run_simulation(wps)
            ts.append(_t.perf_counter() - t0)
            del sc
        out["run_simulation_rays_per_s"] = n / float(np.median(ts[2:]))
        out["run_simulation_ms"] = float(np.median(ts[2:])) * 1e3
        out["run_simulation_ms_min"] = float(np.min(ts[2:])) * 1e3
        if os.environ.get("LRC_BENCH_PROFILE_RUNSIM") == "1":             # where the call's time goes (stderr)
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(10):
                sc = sim.run_simulation(wps)
                del sc
            pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(14)
        sim.raycast_engine.clear_cache()
    except Exception as e:                                               # noqa: BLE001 - a diagnostic figure only
        out["run_simulation_error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-caller-path", action="store_true", help="skip the host-caller timing (N = 1 only)")
    ap.add_argument("--scene", default=SCENE)
    ap.add_argument("--mesh", default=None, metavar="PLY",
                    help="scan this triangle mesh (e.g. an NKSR mesh_dense.ply, as s3dis_simulator.py:556-591 prefers) instead of "
                         "the procedural stand-in; read by the package's own PLY reader; the C3 poses are laid through the "
                         "middle of its bounding box")
    ap.add_argument("--serial", action="store_true",
                    help="N = 1: time lrc_scan_poses_dev + lrc_compact_dev on one stream (the round-3 step) instead of the "
                         "scan pipeline; what the rocprofv3 --stats / PMC passes profile (one un-overlapped trace launch per step)")
    ap.add_argument("--min-seconds", type=float, default=MIN_TIMED_SECONDS,
                    help="repeat the timed block of --steps steps until this much wall time has passed")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="run the N>1 code path (RCCL all-gather, double buffering) with world size 1 and check "
                         "the assembled cloud against the local one")
    ap.add_argument("--virtual-world", type=int, default=1,
                    help="diagnostic, with --dist-selftest only: size the gathered buffers and the cloud rebuild for "
                         "this many ranks (the other ranks' slabs are scanned once before the timed region), to "
                         "measure what the N-GPU step costs a GPU apart from the link transfer")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import lidarcast
    from lidarcast import synth
    from lidarcast._capi import LrcCompactIO
    from lidar import IndoorLidar

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    local_rank %= max(torch.cuda.device_count(), 1)      # tolerate launchers that expose one device per rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.dist_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # "nccl" IS RCCL on ROCm.  LRC_DIST_BACKEND=gloo is a rehearsal aid: several ranks sharing ONE GPU (which RCCL
        # refuses) run the same sharding / gather / rebuild code with the collective staged through the host.
        backend = os.environ.get("LRC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- scene (replicated) and inputs, resident in HBM before the timed region ----
    if args.mesh:
        from lidarcast import ply
        mesh = ply.read_triangle_mesh(args.mesh)
        args.scene = os.path.basename(args.mesh)
    else:
        mesh = synth.make_scene(args.scene)
    ctx = lidarcast.Context(local_rank)
    v32 = np.ascontiguousarray(mesh.vertices, dtype=np.float32)
    f32 = np.ascontiguousarray(mesh.triangles, dtype=np.uint32)
    t0 = time.perf_counter()
    tri_sem, tri_ins = getattr(mesh, "triangle_sem", None), getattr(mesh, "triangle_ins", None)
    scene = lidarcast.Scene(ctx, v32, f32, tri_sem, tri_ins)
    create_first_ms = (time.perf_counter() - t0) * 1e3      # includes the builder's one-time scratch allocation
    info = scene.info
    create_ms = []
    if rank == 0:
        for _ in range(3):                                   # steady state: the next mesh of a batch
            t0 = time.perf_counter()
            again = lidarcast.Scene(ctx, v32, f32, tri_sem, tri_ins)
            create_ms.append((time.perf_counter() - t0) * 1e3)
            again_info = again.info
            again.close()
    sensor = c3_sensor()
    if args.virtual_world > 1 and not (args.dist_selftest and world == 1):
        raise SystemExit("--virtual-world needs --dist-selftest on one GPU")
    job = world if world > 1 else max(args.virtual_world, 1)      # ranks the buffers and the rebuild are sized for
    poses = c3_poses(rank, job)
    if args.mesh:        # the same line of poses, through the middle of this mesh's bounding box at a third of its height
        lo, hi = v32.min(0).astype(np.float64), v32.max(0).astype(np.float64)
        frac = (poses[:, 0, 3] - 1.0) / 3.0
        poses[:, 0, 3] = lo[0] + (0.2 + 0.6 * frac) * (hi[0] - lo[0])
        poses[:, 1, 3] = 0.5 * (lo[1] + hi[1])
        poses[:, 2, 3] = lo[2] + (hi[2] - lo[2]) / 3.0
    dirs = IndoorLidar(intrinsics=sensor, pose=np.eye(4)).sensor_directions()
    P, N = poses.shape[0], dirs.shape[0]
    n = P * N
    d_poses = torch.from_numpy(poses.reshape(P, 16)).to(dev)
    d_dirs = torch.from_numpy(dirs).to(dev)
    dist_path = world > 1 or args.dist_selftest
    want = ("t", "prim", "normal3", "point3", "sem", "ins", "tile_count")
    hits = lidarcast.DeviceHits(n, dev, want=want)
    # N > 1: the own rows of scan i are scattered from its local records while scan i+1 is traced -> two record sets
    hits_sets = [hits, lidarcast.DeviceHits(n, dev, want=want)] if dist_path else [hits]
    # scene cloud: compacted 16-byte rows (x, y, z, sem|ins<<16) in np.vstack order + per-pose counts.
    cloud = torch.empty((n * job, 4), dtype=torch.float32, device=dev)
    counts = torch.zeros(P * job, dtype=torch.int64, device=dev)
    io = LrcCompactIO()
    io.t, io.point3 = hits["t"].data_ptr(), hits["point3"].data_ptr()
    io.sem, io.ins = hits["sem"].data_ptr(), hits["ins"].data_ptr()
    io.tile_count = hits["tile_count"].data_ptr()
    io.counts, io.out_xyzl = counts.data_ptr(), cloud.data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    if dist_path:
        all_poses = np.concatenate([c3_poses(r, job) for r in range(job)]).reshape(P * job, 16)
        d_all_poses = torch.from_numpy(all_poses).to(dev)
    own_slab = rank if world > 1 else 0

    def make_dist_pipe():
        """The `prim` payload through the library's scan pipeline (lrc_pipe_submit_sharded): the trace launches of consecutive
        steps alternate between the pipeline's two streams (nothing between two launches of a stream), the trace writes ids and
        keep counts into the send slab, the all-gather of step k runs on a communication stream behind that trace, and the
        assembly of step k-2 -- rebuild of the other ranks' rows, scatter of the own rows -- rides in the leading workgroups of
        the trace launch of step k instead of waiting, as kernels of its own, for a trace launch to run out of workgroups."""
        from lidarcast.distributed import PrimGather
        st_ = {"i": 0}
        pipe_d = lidarcast.ScanPipe(scene, P, N)
        gathers = [PrimGather(P, N, dist, dev, world=job) for _ in range(2)]
        tickets = [0, 0]
        comm = torch.cuda.Stream(device=dev)
        if world == 1 and job > 1:                 # virtual ranks: their slabs are filled once, outside the timing
            tl = lidarcast.DeviceHits(0, dev, want=())
            for g in gathers:
                for v in range(1, job):
                    tl.struct.prim = g.all_slabs[v * g.words:].data_ptr()
                    tl.struct.tile_count = g.all_slabs[v * g.words + g.n:].data_ptr()
                    scene.scan_poses_dev(d_all_poses[v * P:(v + 1) * P], d_dirs, tl, sensor.max_range, stream)
            torch.cuda.synchronize()

        scanned = [None, None]                     # event: gather + scan of the slabs in gathers[k] are done

        def job_of(k):
            g = gathers[k]
            return lidarcast.ScanPipe.gathered(d_all_poses, g.all_prims, g.all_tile_counts, P, g.stride_bytes, own_slab,
                                               tickets[k], cloud, counts, scan_slot=k)

        def step(timed):
            k = st_["i"] % 2
            st_["i"] += 1
            g = gathers[k]
            main = torch.cuda.current_stream()
            asm = None
            if scanned[k] is not None:
                main.wait_event(scanned[k])      # the collective of step i-2 and the scan of its keep counts are done: its
                asm = job_of(k)                  # send slab (this one) is free again, its assembly rides in this step's launch
            tickets[k] = pipe_d.submit_sharded(d_poses, d_dirs, sensor.max_range, g.prim, g.tile_count, assemble=asm,
                                               stream=main.cuda_stream)
            pipe_d.trace_done(tickets[k], comm.cuda_stream)
            with torch.cuda.stream(comm):
                g.gather(async_op=True)          # ONE RCCL all-gather per scan, ordered behind this step's trace
                g.work.wait()                    # (the communication stream waits for the collective, not the host)
                pipe_d.scan_gathered(d_dirs, job_of(k), comm.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(comm)
            scanned[k] = ev

        def drain():
            main = torch.cuda.current_stream()
            order = [(st_["i"] + j) % 2 for j in range(2)]      # older first
            for k in order:
                if scanned[k] is not None:
                    main.wait_event(scanned[k])
                    pipe_d.assemble(d_dirs, job_of(k), main.cuda_stream)
                    scanned[k] = None
                    gathers[k].work = None
            pipe_d.wait(main.cuda_stream)

        step.pipe = pipe_d
        return step, drain

    def make_dist(payload):
        """The N-rank step for one of the three payloads the package offers (lidarcast.distributed); every one = trace of the
        rank's own poses + ONE all-gather per scan (double buffered: the collective of scan i overlaps the trace of scan
        i+1) + assembly of the whole scene cloud on every GPU:
          prim   4 B per ray: the hit triangle's row (+ per-wave keep counts); the other ranks' rows are rebuilt from the ids
                 (t recomputed with the scan's own ray / triangle test), the own rows scattered from the local records
          range  8 B per ray: (t, label) pairs; every rank rebuilds all rows as o + d * t -- no plane gathers
          rows   16 B per kept ray: locally compacted rows; nothing to rebuild (the close-up copy of the slabs into one
                 array is NOT in this step: a lower bound)
        Returns (step, drain, name)."""
        from lidarcast.distributed import CloudGather, PrimGather, RangeGather
        if payload == "prim_pipe":
            return make_dist_pipe()
        st_ = {"i": 0, "pending": None}
        side = torch.cuda.Stream(device=dev)
        rebuilt = [None, None]                   # event: the assembly that last READ gathers[k]'s receive buffer is done
        if payload == "prim":
            gathers = [PrimGather(P, N, dist, dev, world=job) for _ in range(2)]
        elif payload == "range":
            gathers = [RangeGather(n, dist, dev, world=job) for _ in range(2)]
        else:
            gathers = [CloudGather(n, P, dist, dev, world=job) for _ in range(2)]
        own_ios = []
        for h in hits_sets:                      # the rank's own records, as the own-row scatter of the rebuild reads them
            o = LrcCompactIO()
            o.t, o.point3, o.sem, o.ins = h["t"].data_ptr(), h["point3"].data_ptr(), h["sem"].data_ptr(), h["ins"].data_ptr()
            own_ios.append(o)
        if world == 1 and job > 1 and payload != "rows":
            # virtual ranks (single-GPU diagnostics): their slabs are filled once, outside the timing
            tl = lidarcast.DeviceHits(0, dev, want=())
            for g in gathers:
                for v in range(1, job):
                    if payload == "prim":
                        tl.struct.prim = g.all_slabs[v * g.words:].data_ptr()
                        tl.struct.tile_count = g.all_slabs[v * g.words + g.n:].data_ptr()
                    else:
                        tl.struct.t_label = g.all_pairs[v * n:].data_ptr()
                    scene.scan_poses_dev(d_all_poses[v * P:(v + 1) * P], d_dirs, tl, sensor.max_range, stream)
            torch.cuda.synchronize()

        def assemble(k):
            g = gathers[k]
            with torch.cuda.stream(side):
                g.work.wait()                    # the side stream waits for the collective of that scan
                if payload == "prim":
                    scene.cloud_from_prims_dev(d_all_poses, d_dirs, g.all_prims, cloud, counts, g.all_tile_counts,
                                               poses_per_slab=P, slab_stride_bytes=g.stride_bytes, stream=side.cuda_stream,
                                               own_slab=own_slab, own_io=own_ios[k])
                elif payload == "range":
                    ctx.cloud_from_ranges_dev(d_all_poses, d_dirs, g.all_pairs, cloud, counts, side.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(side)
            rebuilt[k] = ev

        def step(timed):
            timed = timed and len(k_events) < 256       # HIP events around the first few hundred trace launches are plenty
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            k = st_["i"] % 2
            st_["i"] += 1
            g = gathers[k]
            main = torch.cuda.current_stream()
            if g.work is not None:
                g.work.wait()                    # the collective that last read this send slab (scan i-2) is done
            if rebuilt[k] is not None:
                main.wait_event(rebuilt[k])      # ... and so is the assembly that read its receive buffer
            hk = hits_sets[k]
            if payload == "prim":                # the 36-byte record is complete; its id column IS the send slab
                hk.struct.prim, hk.struct.tile_count = g.prim.data_ptr(), g.tile_count.data_ptr()
            elif payload == "range":
                hk.struct.t_label = g.slab.data_ptr()
            scene.scan_poses_dev(d_poses, d_dirs, hk, sensor.max_range, stream)
            if timed:
                e1.record()
                k_events.append((e0, e1))
            if payload == "rows":                # compacted straight into the send slab, per-pose counts in its tail
                o = LrcCompactIO()
                o.t, o.point3, o.sem, o.ins = hk["t"].data_ptr(), hk["point3"].data_ptr(), hk["sem"].data_ptr(), hk["ins"].data_ptr()
                o.tile_count, o.counts, o.out_xyzl = hk["tile_count"].data_ptr(), g.counts.data_ptr(), g.slab.data_ptr()
                ctx.compact_dev(P, N, o, stream)
            g.gather(async_op=True)              # ONE RCCL all-gather per scan, ordered after the trace
            if st_["pending"] is not None:
                assemble(st_["pending"])         # cloud of the previous scan: other stream, beside the next trace
            st_["pending"] = k

        def drain():
            if st_["pending"] is not None:
                assemble(st_["pending"])
                st_["pending"] = None
            torch.cuda.current_stream().wait_stream(side)
            for h in hits_sets:                  # leave the record sets as they were found
                h.struct.prim, h.struct.tile_count = h["prim"].data_ptr(), h["tile_count"].data_ptr()
                h.struct.t_label = None

        return step, drain

    # N = 1 (default): the library's scan pipeline; its rows go to three rotating output buffers (a submit's rows are written
    # up to two submits later).  The bytes are checked against the two-call step below, in this run.
    use_pipe = not dist_path and not args.serial
    if use_pipe:
        pipe = lidarcast.ScanPipe(scene, P, N)
        pipe_rows = [cloud] + [torch.empty_like(cloud) for _ in range(2)]
        pipe_counts = [counts] + [torch.zeros_like(counts) for _ in range(2)]

    k_events = []
    state = {"i": 0}
    dist_payload, dist_calibration = None, None
    if dist_path:
        # Which payload is fastest depends on the link (xGMI all-gather against the rebuild it saves), which only the
        # hardware can say: a few warm steps of each, the fastest runs the timed blocks, all three go into the report.
        def probe(payload, steps=48):
            stp, drn = make_dist(payload)
            for _ in range(8):             # first submits pay one-time set-up (plane table, transposed direction table)
                stp(False)
            drn()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                stp(False)
            drn()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            if world > 1:
                tm = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dt = float(tm.item())
            # every payload is probed from the same state: its buffers go back before the next one allocates (with the earlier
            # steppers' gigabyte of slabs still mapped the later probes ran 5-20 % slower, whichever payload came later)
            del stp, drn
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            return dt * 1e3
        forced = os.environ.get("LRC_DIST_PAYLOAD")
        if forced:
            dist_payload = forced
        else:
            # each payload twice, in one order and then in the reverse one, the better run counts: a probe's figure moves by
            # several percent with what ran before it (profiles/r04_multigpu_virtual_world.txt), whichever payload it is
            order = os.environ.get("LRC_DIST_PROBE_ORDER", "prim_pipe,prim,range,rows").split(",")
            dist_calibration = {pl: probe(pl) for pl in order}
            for pl in reversed(order):
                dist_calibration[pl] = min(dist_calibration[pl], probe(pl))
            # "rows" leaves the close-up copy out, so it must win by more than that copy could cost to be chosen
            # ... and another payload must beat the pipelined one by more than the probes' noise (3 %) to replace it
            dist_payload = min(("prim", "prim_pipe", "range"), key=lambda pl: dist_calibration[pl])
            if dist_calibration[dist_payload] > 0.97 * dist_calibration["prim_pipe"]:
                dist_payload = "prim_pipe"
        dist_step, dist_drain = make_dist(dist_payload)

    def step(timed):
        if use_pipe:
            j = state["i"] % 3
            state["i"] += 1
            pipe.submit(d_poses, d_dirs, sensor.max_range, out_rows_t=pipe_rows[j], counts_t=pipe_counts[j], stream=stream)
            return
        if dist_path:
            dist_step(timed)
            return
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        scene.scan_poses_dev(d_poses, d_dirs, hits, sensor.max_range, stream)
        if timed:
            e1.record()
            k_events.append((e0, e1))
        ctx.compact_dev(P, N, io, stream)

    def drain():
        if use_pipe:
            pipe.wait(stream)          # scatters the rows still in the pipeline, orders this stream behind everything
        if dist_path:
            dist_drain()

    def barrier():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def serial_step(timed):
        """the step as two calls on one stream: trace launch (timed alone by HIP events) + compaction"""
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        scene.scan_poses_dev(d_poses, d_dirs, hits, sensor.max_range, stream)
        if timed:
            e1.record()
            k_events.append((e0, e1))
        ctx.compact_dev(P, N, io, stream)

    serial_ms = None
    if use_pipe:
        # (1) the un-pipelined step on this box: gives the trace launch's own duration (HIP events on its stream; what the
        # roofline is priced with and what rocprofv3 reports for `bench.py --serial`) and the figure the pipeline is held against
        serial_rows, serial_counts = torch.empty_like(cloud), torch.zeros_like(counts)
        io.out_xyzl, io.counts = serial_rows.data_ptr(), serial_counts.data_ptr()
        for _ in range(5):
            serial_step(False)
        torch.cuda.synchronize()
        sb = []
        for _ in range(12):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                serial_step(True)
            torch.cuda.synchronize()
            sb.append((time.perf_counter() - t0) / args.steps)
        serial_ms = float(np.median(sb)) * 1e3

    for _ in range(args.warmup):
        step(False)
    barrier()
    # EXACTLY --steps steps form one timed block, bracketed by barrier + synchronize on both sides.  A block of the
    # default 20 steps lasts ~7 ms, too short for a stable figure (clock ramp) and for the driver's GPU-busy sampler, so
    # the block is repeated until --min-seconds have passed -- half of them before the host legs (CPU baseline, caller
    # path), half after -- and the MEDIAN block is reported; value = rays of one block / its time.
    blocks = []

    def timed_blocks(seconds):
        t_begin = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(True)
            barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                tmax = torch.tensor([dt, time.perf_counter() - t_begin], dtype=torch.float64, device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt, total_elapsed = float(tmax[0].item()), float(tmax[1].item())       # every rank takes the same decision
            else:
                total_elapsed = time.perf_counter() - t_begin
            blocks.append(dt)
            if total_elapsed >= seconds or len(blocks) >= 1 << 16:
                break

    host_legs = rank == 0 and world == 1 and not args.dist_selftest
    timed_blocks(args.min_seconds * (0.5 if host_legs else 1.0))

    if use_pipe:
        # the pipeline's rows are the two-call step's rows, bit for bit (all three rotating buffers)
        kk = int(serial_counts.sum().item())
        for r_, c_ in zip(pipe_rows, pipe_counts):
            assert torch.equal(c_, serial_counts), "scan pipeline: per-pose counts differ from the two-call step"
            assert torch.equal(r_[:kk].view(torch.int32), serial_rows[:kk].view(torch.int32)), \
                "scan pipeline: rows differ from the two-call step"

    if args.dist_selftest and dist_payload != "rows":
        # the cloud rebuilt from the gathered triangle ids / (t, label) pairs must equal the local compaction, bit for bit
        k = int(counts.sum().item())
        rebuilt = cloud[:k].clone()
        rebuilt_counts = counts.clone()
        local = torch.empty((n, 4), dtype=torch.float32, device=dev)
        local_counts = torch.zeros(P, dtype=torch.int64, device=dev)
        io.out_xyzl, io.counts = local.data_ptr(), local_counts.data_ptr()
        row = 0
        for v in range(job):
            scene.scan_poses_dev(d_all_poses[v * P:(v + 1) * P], d_dirs, hits, sensor.max_range, stream)
            ctx.compact_dev(P, N, io, stream)
            torch.cuda.synchronize()
            kv = int(local_counts.sum().item())
            assert torch.equal(rebuilt_counts[v * P:(v + 1) * P], local_counts), f"per-pose counts differ, rank {v}"
            assert torch.equal(rebuilt[row:row + kv].view(torch.int32), local[:kv].view(torch.int32)), \
                f"rebuilt cloud differs in the poses of rank {v}"
            row += kv
        assert row == k, "row totals differ"
        print(f"dist selftest ok: {k} rows rebuilt from the gathered '{dist_payload}' payload == local compaction, "
              f"world {world}, buffers sized for {job} ranks; calibration {dist_calibration}", file=sys.stderr)
    if not k_events:          # the pipelined N-rank step has no un-overlapped launch of its own: time a few here
        io.out_xyzl, io.counts = cloud.data_ptr(), counts.data_ptr()
        for _ in range(12):
            serial_step(True)
        torch.cuda.synchronize()
    kernel_ms = float(np.median([a.elapsed_time(b) for a, b in k_events]))
    hits_total = int(counts.sum().item())
    bpr = bytes_per_ray(info["num_triangles"])
    pmc = pmc_profile("void (anonymous namespace)::trace_kernel<1", n, args.scene)
    kernel_s = kernel_ms * 1e-3
    if pmc is not None:
        # What binds the kernel is vector-ALU issue under divergence, not HBM, so that is the
        # roof it is held to:  achieved = lane-operations that did work per second = SQ_INSTS_VALU x 64 lanes x lane
        # utilisation (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)) / kernel time;  peak = every lane of every
        # SIMD issuing every cycle.  <= 1 by construction (a wave instruction cannot have more than 64 active lanes
        # nor issue faster than one per 2 cycles per SIMD).
        lane_util = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
        lane_ops = pmc["SQ_INSTS_VALU"] * 64.0 * lane_util
        achieved = lane_ops / kernel_s / 1e9
        issue_frac = pmc["SQ_INSTS_VALU"] * 2.0 / (N_SIMDS * CLOCK_GHZ * 1e9 * kernel_s)
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        roofline = {
            "bound": "valu", "achieved": achieved, "peak": VALU_PEAK_GLANEOPS, "unit": "G lane-op/s",
            "frac": achieved / VALU_PEAK_GLANEOPS, "traffic": traffic,
            "valu_issue_frac": issue_frac, "lane_utilisation": lane_util,
            "valu_wave_instructions_per_launch": pmc["SQ_INSTS_VALU"], "active_lane_ops_per_launch": lane_ops,
            # the second, about equally loaded pipe: a wave-wide vector load returns through the CU's 64 B/clk path --
            # 16 clocks for a dwordx4 (the node and triangle fetches), less for narrower loads, so this is an upper bound
            "second_roof": {
                "pipe": "vector-memory return path (64 B/clk/CU)",
                "vmem_rd_wave_instructions_per_launch": pmc.get("SQ_INSTS_VMEM_RD"),
                "busy_frac_upper_bound": (pmc["SQ_INSTS_VMEM_RD"] * 16.0 / (256 * CLOCK_GHZ * 1e9 * kernel_s))
                if "SQ_INSTS_VMEM_RD" in pmc else None,
                "wave_cycles_waiting_frac": (pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"])
                if "SQ_WAIT_ANY" in pmc and pmc.get("SQ_WAVE_CYCLES") else None,
                "salu_per_valu": (pmc["SQ_INSTS_SALU"] / pmc["SQ_INSTS_VALU"]) if "SQ_INSTS_SALU" in pmc else None,
                "note": "SQ_INSTS_VMEM_RD x 16 clk / (256 CU x 2.4 GHz x kernel time): every vector load priced as a "
                        "dwordx4; SQ_WAIT_ANY / SQ_WAVE_CYCLES = share of resident-wave cycles spent waiting",
            },
            "hbm_side": {"traffic_GBps": traffic / kernel_s / 1e9, "frac_of_hbm_peak": traffic / kernel_s / 1e9 / HBM_PEAK_GBS,
                         "note": "2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 correction of MI355X_MICROARCH.md); "
                                 "FETCH counts fabric requests the 256 MiB Infinity Cache mostly serves"},
        }
    else:
        roofline = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_GLANEOPS, "unit": "G lane-op/s", "frac": None,
                    "traffic": None, "note_missing": "no profiles/pmc_latest.json for this binary + workload "
                    "(source fingerprint, kernel, scene or ray count differ): run tools/pmc.sh"}
    roofline.update({
        "kernel": "trace_kernel<GEN=1>", "kernel_ms": kernel_ms, "rays_per_launch": n,
        "kernel_ms_note": ("one un-overlapped trace launch, HIP events on its stream (the serial steps timed before the "
                           "pipelined blocks); `bench.py --serial` under rocprofv3 --stats reports the same launch"
                           if use_pipe else "HIP events on the launch stream over the timed region"),
        "survey_8d_model": {"bytes_per_ray": bpr, "algorithmic_bytes_per_launch": n * bpr,
                            "algorithmic_GBps": n * bpr / kernel_s / 1e9,
                            "note": "SURVEY 8(d) per-ray figure (36 B record + 64 B x ceil(log2(T/4)) descent + 144 B "
                                    "leaf): NOT a lower bound on traffic -- 64 neighbouring rays share their descent "
                                    "through L1/L2/scalar cache -- so it is reported here only, never as frac"},
        "occupancy": dict(scene.occupancy(), max_waves_per_cu=32),
        "note": "achieved = VALU lane-operations that did work per second (counter profile of this binary, "
                "profiles/pmc_latest.json, divided by the kernel time measured live with HIP events on the launch "
                "stream); peak = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz; frac = valu_issue_frac x lane_utilisation",
    })
    caller = caller_path(scene, sensor, poses, dirs, mesh) if (host_legs and not args.no_caller_path) else {}
    cpu = cpu_baseline(mesh, sensor, poses) if (host_legs and not args.no_cpu_baseline) else None
    if host_legs:
        timed_blocks(args.min_seconds * 0.5)          # the second half of the timed blocks, after the host legs
    elapsed = float(np.median(blocks))
    total_rays = n * world * args.steps
    value = total_rays / elapsed
    if use_pipe and pmc is not None:
        # inside the pipeline a launch overlaps its neighbours, so it has no duration of its own; what it costs is at most the
        # step (which also holds the compaction): the VALU work of one launch over ms_per_step is a LOWER bound on what the
        # pipes deliver while the pipeline runs
        step_s = elapsed / args.steps
        roofline["in_pipeline"] = {
            "ms_per_step": step_s * 1e3,
            "frac_lower_bound": lane_ops / step_s / 1e9 / VALU_PEAK_GLANEOPS,
            "valu_issue_frac_lower_bound": pmc["SQ_INSTS_VALU"] * 2.0 / (N_SIMDS * CLOCK_GHZ * 1e9 * step_s),
            "note": "trace-launch VALU work (the isolated launch's counters) / the whole pipelined step, compaction included"}

    if rank == 0:
        res = {
            "metric": "rays/sec (whole node), 32-line x 2048-azimuth sweep",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"C3: create_dense_32line x horizontal_res=2048 ({N} rays/pose) x {P} poses per GPU "
                            f"(straight line, yaw 0) over {args.scene} (procedural stand-in for an S3DIS "
                            f"Area_6 office mesh, 2 cm tessellation, T={info['num_triangles']})",
                **({"virtual_world": job} if world == 1 and job > 1 else {}),
                "rays_per_step_per_gpu": n, "hit_fraction": hits_total / (n * (job if dist_path else 1)),
                "bvh": {"nodes": info["num_nodes"], "depth": info["max_depth"],
                        "builder": "device" if info["device_build"] else "host",
                        "build_ms": round(again_info["build_ms"], 2), "transfer_ms": round(again_info["upload_ms"], 2),
                        "device_MB": round(info["device_bytes"] / 1e6, 1)},
                "scene_create_ms": round(float(np.median(create_ms)), 2),
                "scene_create_first_ms": round(create_first_ms, 2),
                "scene_create_note": "lrc_scene_create: float32 mesh in (pageable) host memory -> validated, BVH + triangle "
                                     "records + quantised node images resident in HBM, built on the GPU "
                                     "(csrc/lrc_bvh_device.hip); median of 3 re-creations, first = with the builder's "
                                     "one-time scratch allocation.  The reference pays an Embree build per POSE",
                "trajectory_including_scene_create_rays_per_s": n / (float(np.median(create_ms)) * 1e-3 + elapsed / args.steps),
                "step_arrangement": ("scan pipeline (lrc_pipe_*): the trace launches of consecutive steps alternate between two "
                                     "internal streams, the rows of step k are scattered by the leading workgroups of the trace "
                                     "launch of step k+2; rows and counts checked in this run against the two-call step, bit for bit"
                                     if use_pipe else
                                     ("two calls on one stream: lrc_scan_poses_dev + lrc_compact_dev (--serial)" if not dist_path
                                      else "one process per GPU: trace + all-gather + cloud assembly, double buffered")),
                **({"gather_payload": dist_payload,
                    "gather_payload_calibration_ms_per_step": dist_calibration,
                    "gather_payload_note": "prim = 4 B triangle id per ray + rebuild of the other ranks' rows; prim_pipe = the same payload "
                                           "through the library's scan pipeline (trace launches overlapped, the assembly in the "
                                           "leading workgroups of a later trace launch); range = 8 B "
                                           "(t, label) per ray + rebuild of all rows without plane gathers; rows = 16 B per kept "
                                           "ray, nothing rebuilt, close-up copy NOT included (lower bound); two runs of 48 warm steps each (the better counts) "
                                           "at start-up, the fastest of prim / prim_pipe / range runs the timed blocks"} if dist_path else {}),
                **({"serial_ms_per_step": serial_ms,
                    "serial_note": "the same step as lrc_scan_poses_dev + lrc_compact_dev on one stream, same box, same run "
                                   "(median of 12 blocks); roofline.kernel_ms is the trace launch of THESE steps, alone on the "
                                   "chip"} if serial_ms is not None else {}),
                "caller_path_note": "value is the device-resident loop; caller_path_rays_per_s is what a host caller of "
                                    "the plugin surface gets (kept rows in page-locked host memory, PCIe included)",
                "step": "in-kernel ray generation + BVH traversal + hit write-back (36 B/ray) + "
                        + ("stable compaction into the scene cloud (16 B/hit)" if world == 1 else
                           "one RCCL all-gather of the hit triangle ids (4 B/ray + 4 B per 64 rays of keep counts) + "
                           "assembly of the whole scene cloud (16 B/hit) on every GPU: the other ranks' rows rebuilt "
                           "from their ids, the own rows scattered from the local records"),
            },
            "roofline": roofline,
        }
        res["config"].update(caller)
        res["timed"] = {"blocks": len(blocks), "block_steps": args.steps, "block_ms_median": elapsed * 1e3,
                        "block_ms_min": min(blocks) * 1e3, "block_ms_max": max(blocks) * 1e3,
                        "note": "value = rays of one block of --steps steps / the median block time"}
        if cpu is not None:
            res["cpu_baseline"] = cpu
            res["cpu_baseline"]["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(res))
    if world > 1 or args.dist_selftest:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Drop-in caller of the engines: the reference's S3DISSimulator scan stage
(reference: s3dis_simulator.py:36-77 construction, :220-296 run_simulation).

Scope: sensor + engine selection, trajectory (planned or straight), the scan loop and the result files, under the
reference's method names and signatures.  Scene loading takes a mesh object (anything with ``.vertices`` /
``.triangles``) or a PLY path; visualisation, furniture-aware collision checking, the batch / CLI drivers and NKSR
reconstruction are not part of this package.

``run_simulation`` keeps the reference's per-frame ScanQuality formulas, including two quirks of the
reference loop that a drop-in must reproduce to give the same statistics (``bug_compatible=True``):
the incident angles are overwritten with zeros (:266-269) and the range statistics are norms from the
WORLD origin, not from the sensor (:283-284).
"""
import itertools
import json
import time
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from containers import RoomBounds, S3DISScene, S3DISSimFrame, S3DISSimScene, ScanQuality
from lidar import (DualAxisLidarIntrinsics, Indoor8LineLidarIntrinsics, create_lidar)
from raycast_engine import RaycastEngineCPU, RaycastEngineGPU
from trajectory import AutoTrajectoryGenerator, PathType, SmartTrajectoryGenerator, Waypoint, poses_from_waypoints


_POOL = None             # thread pool of _quality_many, created on first use
_NO_GROUP = object()     # run_simulation(process_group=_NO_GROUP): single-process scan even inside a distributed job


class S3DISSimulator:
    def __init__(self, config: Dict[str, Any], use_dense_lidar: bool = False, use_blk2go: bool = False,
                 bug_compatible: bool = True):
        self.config = config
        self.use_dense_lidar = use_dense_lidar
        self.use_blk2go = use_blk2go
        self.bug_compatible = bug_compatible
        self.scene: Optional[S3DISScene] = None
        self.lidar_config = None
        self.raycast_engine = None
        self.trajectory_generator: Optional[SmartTrajectoryGenerator] = None
        self.auto_trajectory_generator: Optional[AutoTrajectoryGenerator] = None
        self.collision_detector = None          # furniture-aware planning is out of scope
        self._initialize_components()

    def _initialize_components(self):
        if self.use_blk2go:
            self.lidar_config = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
        elif self.use_dense_lidar:
            self.lidar_config = Indoor8LineLidarIntrinsics.create_dense_32line()
        else:
            self.lidar_config = Indoor8LineLidarIntrinsics.create_standard_8line()
        # Both names are the HIP engine in this package; construction raises without a GPU.
        use_gpu = self.config.get("raycast_engine", {}).get("use_gpu", False)
        self.raycast_engine = RaycastEngineGPU() if use_gpu else RaycastEngineCPU()

    def load_scene(self, scene_path, scene_name: Optional[str] = None) -> S3DISScene:
        """scene_path: path to a triangle-mesh PLY (name defaults to the file stem, reference :78-129), or a mesh
        object."""
        if isinstance(scene_path, (str, bytes)) or hasattr(scene_path, "__fspath__"):
            from lidarcast.ply import read_triangle_mesh
            mesh = read_triangle_mesh(scene_path)
            scene_name = scene_name or Path(str(scene_path)).stem
            if len(np.asarray(mesh.vertices)) == 0:
                raise ValueError(f"Failed to load mesh file: {scene_path}")
        else:
            mesh = scene_path
            if len(np.asarray(mesh.vertices)) == 0:
                raise ValueError("Failed to load mesh: no vertices")
        self.scene = S3DISScene(scene_name or "scene", mesh, RoomBounds.from_vertices(mesh.vertices))
        self.trajectory_generator = SmartTrajectoryGenerator(
            self._bounds_dict(), robot_height=self.config.get("trajectory", {}).get("robot_height", 1.0))
        # same planner settings as the reference (s3dis_simulator.py:127): reduced radius for narrow spaces
        self.auto_trajectory_generator = AutoTrajectoryGenerator(robot_radius=0.15, context=self.raycast_engine.ctx)
        return self.scene

    def _bounds_dict(self) -> Dict[str, float]:
        b = self.scene.room_bounds
        return {"x_min": b.x_min, "x_max": b.x_max, "y_min": b.y_min, "y_max": b.y_max,
                "z_min": b.z_min, "z_max": b.z_max}

    def generate_auto_trajectory(self, num_waypoints: int = 20) -> Tuple[List[Waypoint], Dict[str, Any]]:
        """Planned trajectory for the loaded scene (reference s3dis_simulator.py:132-167)."""
        if self.auto_trajectory_generator is None or self.scene is None:
            raise ValueError("Scene not loaded. Call load_scene() first.")
        return self.auto_trajectory_generator.generate_optimal_trajectory(
            mesh=self.scene.room_mesh, room_bounds=self._bounds_dict(), num_waypoints=num_waypoints)

    def generate_trajectory(self, start_point: Tuple[float, float, float], end_point: Tuple[float, float, float],
                            path_type: PathType = PathType.STRAIGHT,
                            num_waypoints: int = 20) -> Tuple[List[Waypoint], Dict[str, Any]]:
        """Trajectory between two positions -> (waypoints, quality as a dict) (reference :181-218; the furniture
        re-planning branch has nothing to act on here)."""
        if self.trajectory_generator is None:
            raise ValueError("Scene not loaded. Call load_scene() first.")
        waypoints, quality = self.trajectory_generator.generate_trajectory(
            start_point=start_point, end_point=end_point, path_type=path_type, num_waypoints=num_waypoints)
        return waypoints, quality.to_dict()

    def add_furniture(self, furniture_mesh, name: str, category: str = "unknown"):
        """The reference forwards this to its CollisionDetector, which does not import; furniture
        that is part of the room mesh occludes rays without it."""
        raise NotImplementedError("furniture-aware collision checking is out of scope (DESIGN.md section 9); "
                                  "put the furniture into the room mesh")

    def save_results(self, sim_scene: S3DISSimScene, output_dir, waypoints: Optional[List[Waypoint]] = None,
                     save_visualizations: bool = True):
        """Result files of the scene (statistics, summary, both combined clouds; reference :298-371).  The
        visualisations the reference adds are out of scope; the flag is accepted and ignored."""
        sim_scene.save_results(Path(output_dir))

    def run_complete_simulation(self, scene_path, start_point: Tuple[float, float, float],
                                end_point: Tuple[float, float, float], path_type: PathType = PathType.STRAIGHT,
                                num_waypoints: int = 20, output_dir=None,
                                scene_name: Optional[str] = None) -> S3DISSimScene:
        """load_scene -> generate_trajectory -> run_simulation -> save_results (reference :373-413)."""
        self.load_scene(scene_path, scene_name)
        waypoints, _ = self.generate_trajectory(start_point, end_point, path_type, num_waypoints)
        sim_scene = self.run_simulation(waypoints)
        self.save_results(sim_scene, Path("s3dis_simulation_results") if output_dir is None else Path(output_dir),
                          waypoints)
        return sim_scene

    def run_auto_simulation(self, scene_path, num_waypoints: int = 20, output_dir=None,
                            scene_name: Optional[str] = None) -> S3DISSimScene:
        """load_scene -> generate_auto_trajectory -> run_simulation -> save_results + trajectory_analysis.json
        (reference :415-455).  The planned waypoints and the planner's report stay available as
        ``last_waypoints`` / ``last_analysis``."""
        self.load_scene(scene_path, scene_name)
        waypoints, analysis = self.generate_auto_trajectory(num_waypoints)
        self.last_waypoints, self.last_analysis = waypoints, analysis
        sim_scene = self.run_simulation(waypoints)
        out = Path("s3dis_auto_simulation_results") if output_dir is None else Path(output_dir)
        self.save_results(sim_scene, out, waypoints)
        with open(out / "trajectory_analysis.json", "w", encoding="utf-8") as f:
            json.dump(analysis, f, indent=2, ensure_ascii=False)
        return sim_scene

    # ---- the scan stage ---------------------------------------------------------------------------
    def _quality(self, points, incident_angles, total_points_per_scan, room_volume, ranges=None) -> ScanQuality:
        k = len(points)
        if ranges is None:
            ranges = np.linalg.norm(points, axis=1) if k > 0 else None     # from the world origin, as the reference
        return ScanQuality(
            coverage_ratio=k / total_points_per_scan, num_points=k,
            incident_angle_mean=np.mean(incident_angles) if len(incident_angles) > 0 else 0,
            incident_angle_std=np.std(incident_angles) if len(incident_angles) > 0 else 0,
            scan_density=k / room_volume,
            range_mean=np.mean(ranges) if k > 0 else 0, range_std=np.std(ranges) if k > 0 else 0)

    def _quality_from_stats(self, fr, total_points_per_scan, room_volume):
        """ScanQuality of every frame from the per-pose statistics the device computed with numpy's arithmetic."""
        counts = fr["counts"].tolist()
        zero = np.float64(0.0)
        P = len(counts)
        am = [zero] * P if self.bug_compatible else list(fr["incident_mean"])
        asd = [zero] * P if self.bug_compatible else list(fr["incident_std"])
        rm, rs = list(fr["range_origin_mean"]), list(fr["range_origin_std"])      # numpy scalars, as np.mean returns them
        out = []
        for i, k in enumerate(counts):
            if k == 0:
                out.append(ScanQuality(0.0, 0, 0, 0, 0.0, 0, 0))
            else:
                out.append(ScanQuality(k / total_points_per_scan, k, am[i], asd[i], k / room_volume, rm[i], rs[i]))
        return out

    def _quality_many(self, points, angles, total_points_per_scan, room_volume, ranges):
        """ScanQuality of every frame; frames are independent, so a small thread pool reduces them side by side."""
        zero_angles = self.bug_compatible      # mean and std of an all-zero block are exactly 0.0: no need to reduce it

        def one(i):
            k = len(points[i])
            r = ranges[i]
            a = angles[i]
            return ScanQuality(
                coverage_ratio=k / total_points_per_scan, num_points=k,
                incident_angle_mean=(np.float64(0.0) if zero_angles else np.mean(a)) if k > 0 else 0,
                incident_angle_std=(np.float64(0.0) if zero_angles else np.std(a)) if k > 0 else 0,
                scan_density=k / room_volume,
                range_mean=np.mean(r) if k > 0 else 0, range_std=np.std(r) if k > 0 else 0)
        n = len(points)
        if n < 8 or sum(len(p) for p in points) < (1 << 18):
            return [one(i) for i in range(n)]
        global _POOL
        if _POOL is None:
            import os
            from concurrent.futures import ThreadPoolExecutor
            _POOL = ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 2))))
        return list(_POOL.map(one, range(n)))

    def run_simulation(self, waypoints: List[Waypoint], process_group=None) -> S3DISSimScene:
        """The scan stage (reference :220-296).  Inside an initialised ``torch.distributed`` job with more than one
        rank (one process per GPU, backend "nccl" = RCCL), or with an explicit ``process_group``, the waypoints are
        sharded over the ranks in contiguous blocks, ONE all-gather per scan assembles the scene, and every rank
        returns the complete S3DISSimScene -- identical to the single-process result."""
        if self.scene is None:
            raise ValueError("Scene not loaded. Call load_scene() first.")
        if self.raycast_engine is None:
            raise ValueError("Raycast engine is not initialized.")
        mesh = self.scene.room_mesh
        sim_scene = S3DISSimScene(scene_name=self.scene.scene_name, simulation_config=self.config, mesh=mesh,
                                  s3dis_data_root=self.config.get("s3dis_data_root"),
                                  area=self.config.get("area"), room=self.config.get("room"))
        start = time.time()
        total = self.lidar_config.get_total_points_per_scan()
        volume = self.scene.room_bounds.get_volume()

        # The whole trajectory in ONE launch; compaction into per-pose frames happens in HBM and only the kept rows
        # cross PCIe, into page-locked buffers -- the frames below are views of those (lrc_scan_poses_compact).
        # In bug-compatible mode the incident angles are overwritten with zeros anyway (reference :266-269), so they
        # are not even transferred.  range_origin is |point| from the WORLD origin (reference :283-284), float32,
        # formed on the device exactly as np.linalg.norm(points, axis=1) forms it.
        engine = self.raycast_engine
        # the per-frame mean / std of the ScanQuality records come from the device as well, computed with numpy's own
        # summation order (csrc/lrc_stats.h), so neither the range column nor a host reduction is needed
        from lidarcast.npmodel import reductions_match
        device_stats = reductions_match()       # a numpy that sums differently reduces the columns itself
        batched = isinstance(self.lidar_config, Indoor8LineLidarIntrinsics) and \
            self.lidar_config.vertical_degrees is not None
        # the hit triangles' labels are an extra of this package's frames (the reference's frame has none,
        # containers/s3dis_sim_frame.py:90-101): by default they stay on the device side -- a second, labels-only scan of
        # the same poses fetches them the first time a frame's labels are looked at (or at export) -- so that a caller who
        # never does is not made to wait for four of every sixteen bytes to cross PCIe.  raycast_engine.eager_labels: true
        # brings them with the points as before.
        lazy_labels = batched and not bool(self.config.get("raycast_engine", {}).get("eager_labels", False))
        want = ("point3",) + (() if lazy_labels else ("sem", "ins")) + \
            ("range_origin_stats" if device_stats else "range_origin",) + \
            (() if self.bug_compatible else (("incident_deg", "incident_stats") if device_stats else ("incident_deg",)))
        device_gen = bool(self.config.get("raycast_engine", {}).get("device_ray_generation", False))
        fr = None
        from lidarcast.distributed import active_group, scan_frames_sharded, scan_lidars_sharded
        dist, group = (None, None) if process_group is _NO_GROUP else active_group(process_group)
        if dist is not None and len(waypoints) > 0:
            if batched:
                fr = scan_frames_sharded(engine, self.lidar_config, poses_from_waypoints(waypoints), mesh, dist, group)
            else:
                lidars = [create_lidar(self.lidar_config, wp.to_pose_matrix()) for wp in waypoints]
                fr = scan_lidars_sharded(engine, lidars, mesh, dist, group)
            # attributes the gathered rows do not carry are the reference's own numpy expressions of the points; the
            # range statistics come from the device when the engine computed them on the assembled rows
            from lidarcast.npmodel import reductions_match
            if "range_origin_mean" in fr and not reductions_match():
                del fr["range_origin_mean"], fr["range_origin_std"]
            if "range_origin_mean" not in fr:
                fr["range_origin"] = np.linalg.norm(fr["point3"], axis=1) if fr["total"] else np.zeros(0, np.float32)
            if not self.bug_compatible:
                fr.pop("range_origin_mean", None), fr.pop("range_origin_std", None)     # angle statistics: host numpy
                if "range_origin" not in fr:
                    fr["range_origin"] = np.linalg.norm(fr["point3"], axis=1) if fr["total"] else np.zeros(0, np.float32)
                cen = np.repeat(np.stack([wp.to_pose_matrix()[:3, 3] for wp in waypoints]), fr["counts"], axis=0)
                v = fr["point3"] - cen
                v = v / np.linalg.norm(v, axis=1, keepdims=True)
                fr["incident_deg"] = np.degrees(np.arccos(np.abs(v[:, 2])))      # raycast_engine_cpu.py:100-107
        elif len(waypoints) > 0 and batched:
            fr = engine.scan_frames(self.lidar_config, poses_from_waypoints(waypoints), mesh, want=want)
        elif len(waypoints) > 0 and device_gen and isinstance(self.lidar_config, DualAxisLidarIntrinsics):
            # opt-in: dual-axis rays generated in the kernel from the host-drawn scan angles
            lidars = [create_lidar(self.lidar_config, wp.to_pose_matrix()) for wp in waypoints]
            fr = engine.scan_frames_dual_axis(lidars, mesh, want=want)
        elif len(waypoints) > 0 and isinstance(self.lidar_config, DualAxisLidarIntrinsics) and \
                hasattr(engine, "scan_frames_lidars"):
            # host-generated rays (dual-axis sensor, the bit-exact default): all rays of all poses at a fixed stride with
            # the dropout mask, one launch, compaction and statistics in HBM
            lidars = [create_lidar(self.lidar_config, wp.to_pose_matrix()) for wp in waypoints]
            fr = engine.scan_frames_lidars(lidars, mesh, want=want)
        elif len(waypoints) > 0:
            # any other sensor object with get_rays(): all poses in one launch, ragged segments, masked on the host
            lidars = [create_lidar(self.lidar_config, wp.to_pose_matrix()) for wp in waypoints]
            seg, off = engine.scan_lidars(lidars, mesh, want=("t", "point3", "incident_deg", "sem", "ins"))
        if fr is not None:
            if self.bug_compatible:            # one zero block, frames take views of it (reference :266-269)
                fr["incident_deg"] = np.zeros(fr["total"])
            counts_l = fr["counts"].tolist()
            ends_l = list(itertools.accumulate(counts_l))
            pts_f, ang_f = ([fr[a][e - c:e] for c, e in zip(counts_l, ends_l)] for a in ("point3", "incident_deg"))
            if "sem" in fr:
                sem_f, ins_f = ([fr[a][e - c:e] for c, e in zip(counts_l, ends_l)] for a in ("sem", "ins"))
                src_f = itertools.repeat(None)
            else:      # labels on demand: one labels-only scan of the trajectory, shared by its frames
                sem_f = ins_f = itertools.repeat(None)
                src_f = itertools.repeat(_LazyTrajectoryLabels(engine, self.lidar_config, poses_from_waypoints(waypoints),
                                                               mesh, counts_l))
            if "range_origin_mean" in fr:      # statistics from the device
                qual = self._quality_from_stats(fr, total, volume)
            else:
                # (sharded scans) per-frame statistics are numpy reductions over 10^4..10^5 values each; they release
                # the GIL, so the frames of a long trajectory are reduced side by side (same numpy calls, same values)
                qual = self._quality_many(pts_f, ang_f, total, volume, engine.split_frames(fr, "range_origin"))
        if fr is not None:
            sim_scene.frames.extend(map(S3DISSimFrame, range(len(waypoints)), pts_f, ang_f, qual, itertools.repeat(None),
                                        sem_f, ins_f, src_f))
        for i, wp in enumerate(waypoints if fr is None else ()):
            a, b = off[i], off[i + 1]
            keep = seg["t"][a:b] != np.inf
            points, angles = seg["point3"][a:b][keep], seg["incident_deg"][a:b][keep]
            sem, ins = seg["sem"][a:b][keep], seg["ins"][a:b][keep]
            if self.bug_compatible:
                angles = np.zeros(len(points))                # reference :266-269
            q = self._quality(points, angles, total, volume)
            sim_scene.append_frame(S3DISSimFrame(i, points, angles, q, semantic_labels=sem,
                                                 instance_labels=ins))
        sim_scene.compute_statistics(time.time() - start)
        return sim_scene


class _LazyTrajectoryLabels:
    """The hit triangles' labels of a scanned trajectory, fetched when a frame is first asked for them: one labels-only scan
    of the same poses over the same mesh (the scan is a pure function of both: same kept rays, same order), whose result
    all frames of the trajectory share."""

    def __init__(self, engine, intrinsics, poses, mesh, counts):
        self._args = (engine, intrinsics, np.array(poses, dtype=np.float64, copy=True), mesh)
        self._counts = list(counts)
        self._sem = self._ins = None
        import threading
        self._lock = threading.Lock()

    def frame_labels(self, i):
        with self._lock:
            if self._sem is None:
                engine, intrinsics, poses, mesh = self._args
                fr = engine.scan_frames(intrinsics, poses, mesh, want=("sem", "ins"))
                if fr["counts"].tolist() != self._counts:
                    raise RuntimeError("the mesh or the scan options changed between the scan and the first look at its labels")
                self._sem, self._ins, self._ends = fr["sem"], fr["ins"], list(itertools.accumulate(self._counts))
                self._args = None
        e, c = self._ends[i], self._counts[i]
        return self._sem[e - c:e], self._ins[e - c:e]


def load_config(config_path: str) -> Dict[str, Any]:
    """YAML configuration file -> dict (reference :458-465)."""
    import yaml
    with open(config_path, "r", encoding="utf-8") as f:
        return yaml.safe_load(f)


def load_default_config() -> Dict[str, Any]:
    """configs/default_config.yaml next to this module (the reference looks for the same file and does not ship
    it either: FileNotFoundError there and here)."""
    return load_config(str(Path(__file__).parent / "configs" / "default_config.yaml"))


def create_simulator_from_config(config_path: Optional[str] = None) -> S3DISSimulator:
    return S3DISSimulator(load_default_config() if config_path is None else load_config(config_path))


def run_scene_batch(scenes, trajectories, sensor=None, config: Optional[Dict[str, Any]] = None,
                    use_dense_lidar: bool = False, use_blk2go: bool = False, output_base_dir=None,
                    process_group=None, skip_existing: bool = True, on_scene=None) -> Dict[str, Any]:
    """Several scenes through one simulator, the job of the reference's batch loop (s3dis_simulator.py:594-726:
    per scene load -> trajectory -> run_simulation -> save_results, failures collected, finished scenes skipped) as
    a function instead of a hard-coded ``main``.

    scenes        list of (name, mesh object or PLY path)
    trajectories  {name: waypoints (List[Waypoint]) or (P,4,4) poses}, or an int = waypoints planned per scene
    sensor        optional intrinsics record replacing the simulator's default (e.g. a 32-line x 2048 sweep)
    output_base_dir  results of scene ``name`` go to <dir>/<name>; a scene whose labelled cloud and statistics
                  file exist is skipped (the reference's resume rule, :637-648)
    on_scene      optional callable (name, sim_scene): called when a scene is done (after its files are written); the scene
                  is then NOT kept in the report, as the reference's loop keeps none (:661-700) -- its frames' page-locked
                  buffers go back to the pool and the next scene's scan reuses them instead of locking fresh pages
    Inside a torch.distributed job the scenes are dealt round-robin to the ranks (each scene is one rank's work, its
    trajectory is not sharded again) and the per-scene summaries are gathered; ``sim_scene`` objects stay on the
    rank that produced them.  Returns {"scenes": {name: {...}}, "failed": [...], "skipped": [...], "total_rays",
    "seconds", "rays_per_s", "build_seconds", "seconds_including_build", "rays_per_s_including_build"}: ``seconds`` = the
    scan stages only, ``build_seconds`` = the scene builds (mesh -> BVH resident in HBM, which the reference pays per
    POSE, raycast_engine_cpu.py:46-47, and this engine once per mesh, on the GPU); file writing is in neither."""
    from lidarcast.distributed import active_group
    dist, group = active_group(process_group)
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist is not None else (0, 1)
    sim = S3DISSimulator(config or {"raycast_engine": {"use_gpu": True}}, use_dense_lidar=use_dense_lidar,
                         use_blk2go=use_blk2go)
    if sensor is not None:
        sim.lidar_config = sensor
    done, failed, skipped = {}, [], []
    rays_per_pose = sim.lidar_config.get_total_points_per_scan()
    for i, (name, source) in enumerate(scenes):
        if i % world != rank:
            continue
        out_dir = None if output_base_dir is None else Path(output_base_dir) / name
        if (skip_existing and out_dir is not None and (out_dir / "combined_pointcloud_with_label.ply").exists()
                and (out_dir / "simulation_statistics.txt").exists()):
            skipped.append(name)
            continue
        try:
            sim.load_scene(source, name)
            traj = trajectories if isinstance(trajectories, int) else trajectories[name]
            if isinstance(traj, int):
                waypoints, _ = sim.generate_auto_trajectory(traj)
            elif isinstance(traj, np.ndarray):
                waypoints = [Waypoint(m[0, 3], m[1, 3], m[2, 3], yaw=float(np.arctan2(m[1, 0], m[0, 0])),
                                      timestamp=float(k)) for k, m in enumerate(traj.reshape(-1, 4, 4))]
            else:
                waypoints = list(traj)
            tb = time.perf_counter()
            sim.raycast_engine.scene_for(sim.scene.room_mesh)        # scene build: timed on its own
            t0 = time.perf_counter()
            sim_scene = sim.run_simulation(waypoints, process_group=_NO_GROUP)
            dt = time.perf_counter() - t0
            if out_dir is not None:
                sim.save_results(sim_scene, out_dir, waypoints)
            done[name] = {"sim_scene": sim_scene, "frames": len(sim_scene.frames), "rays": len(waypoints) * rays_per_pose,
                          "points": int(sim_scene.get_total_points()), "seconds": dt, "build_seconds": t0 - tb}
            if on_scene is not None:
                on_scene(name, sim_scene)
                done[name]["sim_scene"] = None
            del sim_scene
        except Exception as e:                                       # noqa: BLE001 - the reference collects and goes on
            failed.append((name, str(e)))
    if dist is not None:
        parts = [None] * world
        summary = {k: {a: b for a, b in v.items() if a != "sim_scene"} for k, v in done.items()}
        dist.all_gather_object(parts, (summary, failed, skipped), group=group)
        for r, (sm, fl, sk) in enumerate(parts):
            if r != rank:
                done.update(sm)
                failed += fl
                skipped += sk
        seconds = max(sum(v["seconds"] for v in part[0].values()) for part in parts)    # ranks work side by side
        with_build = max(sum(v["seconds"] + v["build_seconds"] for v in part[0].values()) for part in parts)
    else:
        seconds = sum(v["seconds"] for v in done.values())
        with_build = sum(v["seconds"] + v["build_seconds"] for v in done.values())
    total_rays = sum(v["rays"] for v in done.values())
    order = [n for n, _ in scenes]
    return {"scenes": {n: done[n] for n in order if n in done}, "failed": failed, "skipped": skipped, "ranks": world,
            "total_rays": int(total_rays), "seconds": seconds, "rays_per_s": total_rays / seconds if seconds > 0 else 0.0,
            "build_seconds": with_build - seconds, "seconds_including_build": with_build,
            "rays_per_s_including_build": total_rays / with_build if with_build > 0 else 0.0}

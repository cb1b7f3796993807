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
import json
import time
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from containers import RoomBounds, S3DISScene, S3DISSimFrame, S3DISSimScene, ScanQuality
from lidar import (DualAxisLidarIntrinsics, Indoor8LineLidarIntrinsics, create_lidar)
from raycast_engine import RaycastEngineCPU, RaycastEngineGPU
from trajectory import AutoTrajectoryGenerator, PathType, SmartTrajectoryGenerator, Waypoint, poses_from_waypoints


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
    def _quality(self, points, incident_angles, total_points_per_scan, room_volume) -> ScanQuality:
        k = len(points)
        ranges = np.linalg.norm(points, axis=1) if k > 0 else None     # from the world origin, as the reference
        return ScanQuality(
            coverage_ratio=k / total_points_per_scan, num_points=k,
            incident_angle_mean=np.mean(incident_angles) if len(incident_angles) > 0 else 0,
            incident_angle_std=np.std(incident_angles) if len(incident_angles) > 0 else 0,
            scan_density=k / room_volume,
            range_mean=np.mean(ranges) if k > 0 else 0, range_std=np.std(ranges) if k > 0 else 0)

    def run_simulation(self, waypoints: List[Waypoint]) -> S3DISSimScene:
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

        batched = isinstance(self.lidar_config, Indoor8LineLidarIntrinsics) and \
            self.lidar_config.vertical_degrees is not None and len(waypoints) > 0
        if batched:
            # every pose in one launch; rays generated in the kernel
            rec, n = self.raycast_engine.scan_poses(self.lidar_config, poses_from_waypoints(waypoints), mesh,
                                                    want=("t", "point3", "incident_deg", "sem", "ins"))
        elif len(waypoints) > 0:
            # host-generated rays (dual-axis sensor): all poses in one launch, ragged segments
            lidars = [create_lidar(self.lidar_config, wp.to_pose_matrix()) for wp in waypoints]
            seg, off = self.raycast_engine.scan_lidars(lidars, mesh, want=("t", "point3", "incident_deg", "sem", "ins"))
        for i, wp in enumerate(waypoints):
            if batched:
                keep = rec["t"][i] != np.inf
                points, angles = rec["point3"][i][keep], rec["incident_deg"][i][keep]
                sem, ins = rec["sem"][i][keep], rec["ins"][i][keep]
            else:
                a, b = off[i], off[i + 1]
                keep = seg["t"][a:b] != np.inf
                points, angles = seg["point3"][a:b][keep], seg["incident_deg"][a:b][keep]
                sem, ins = seg["sem"][a:b][keep], seg["ins"][a:b][keep]
            if self.bug_compatible:
                angles = np.zeros(len(points))            # reference :266-269
            q = self._quality(points, angles, total, volume)
            sim_scene.append_frame(S3DISSimFrame(i, points, angles, q, semantic_labels=sem,
                                                 instance_labels=ins))
        sim_scene.compute_statistics(time.time() - start)
        return sim_scene


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

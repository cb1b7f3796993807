#!/usr/bin/env python3
"""Row N2 measurement: the planner on the C3 scene (synth_A6_office2, 308 k vertices) and the whole auto pipeline
(plan -> 128-pose scan); beside it the reference's formulation of the robot-cube test (one numpy pass over all
vertices per position, trajectory/auto_trajectory_generator.py:219-238) timed on a sample of the same positions."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401
import numpy as np  # noqa: E402
import s3dis_simulator  # noqa: E402
from lidarcast import synth  # noqa: E402

mesh = synth.make_scene(bench.SCENE)
sim = s3dis_simulator.S3DISSimulator({"raycast_engine": {"use_gpu": True}}, use_dense_lidar=True)
sim.load_scene(mesh, "synth_A6_office2")
np.random.seed(0)
t0 = time.perf_counter()
wps, info = sim.generate_auto_trajectory(num_waypoints=64)
plan_s = time.perf_counter() - t0
gen = sim.auto_trajectory_generator
n_grid = len(gen.room_analysis.free_space_points) + len(gen.room_analysis.obstacle_points)
n_positions = n_grid + info["total_candidates"] * len(wps)

v = np.asarray(mesh.vertices)
r = gen.robot_radius
sample = np.array(gen.room_analysis.free_space_points[:40])
t0 = time.perf_counter()
for p in sample:
    lo, hi = p - r, p + r
    np.any((v[:, 0] >= lo[0]) & (v[:, 0] <= hi[0]) & (v[:, 1] >= lo[1]) & (v[:, 1] <= hi[1]) &
           (v[:, 2] >= lo[2]) & (v[:, 2] <= hi[2]))
per_pos = (time.perf_counter() - t0) / len(sample)
occ = gen._occupancy(mesh)
pts = np.array(gen.room_analysis.free_space_points)
t0 = time.perf_counter()
occ.occupied(pts, r)
gpu_grid_s = time.perf_counter() - t0

sim2 = s3dis_simulator.S3DISSimulator({"raycast_engine": {"use_gpu": True}}, use_dense_lidar=True)
np.random.seed(0)
t0 = time.perf_counter()
sim2.load_scene(mesh)                                   # run_auto_simulation without its result files
wps2, _ = sim2.generate_auto_trajectory(64)
scene = sim2.run_simulation(wps2)
auto_s = time.perf_counter() - t0
print(json.dumps({
    "vertices": int(len(v)), "grid_positions": n_grid, "candidates": info["total_candidates"],
    "waypoints": len(wps), "cube_tests_total": n_positions,
    "planner_seconds": plan_s, "gpu_cube_test_seconds_for_grid": gpu_grid_s,
    "numpy_per_position_seconds": per_pos, "numpy_cube_tests_extrapolated_seconds": per_pos * n_positions,
    "run_auto_simulation_seconds_incl_scene_build": auto_s, "frames": scene.get_total_frames(),
    "points": scene.get_total_points(),
    "best": {k: float(val) if not isinstance(val, list) else val for k, val in info["best_trajectory"].items()},
}, indent=1))

#!/usr/bin/env python3
"""tools/rng_time.py [poses] -- the seeded stream of the BLK2GO sensor (per pose 128 000 normals + 64 000 uniforms):
numpy's own draws against the native restatement (lidarcast.nprandom.scan_draws, csrc/lrc_nprandom.cpp), and the whole
host ray generation of a trajectory (draws + trigonometry + rotation, raycast_engine_hip.dual_axis_rays_batch)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401
import numpy as np  # noqa: E402
from lidarcast import nprandom  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
np.random.seed(0)
t0 = time.perf_counter()
for _ in range(32):
    np.random.normal(0, 1e-3, size=128000)
    np.random.random(64000)
t_np = (time.perf_counter() - t0) / 32
print(f"numpy: {t_np * 1e3:.3f} ms per pose -> {t_np * P:.3f} s for {P} poses (sequential by construction)")
nprandom.scan_draws(2, 128000, 64000, 0.0, 1e-3)
for th in (1, 2, 4, 8, 16):
    np.random.seed(0)
    t0 = time.perf_counter()
    for a in range(0, P, 16):
        nprandom.scan_draws(min(16, P - a), 128000, 64000, 0.0, 1e-3, threads=th)
    dt = time.perf_counter() - t0
    print(f"native, {th:2d} threads, runs of 16 poses: {dt / P * 1e3:.3f} ms per pose -> {dt:.3f} s for {P} poses")

for run in (16, 32, 64, 128, 256):
    np.random.seed(0)
    t0 = time.perf_counter()
    for a in range(0, P, run):
        nprandom.scan_draws(min(run, P - a), 128000, 64000, 0.0, 1e-3)
    dt = time.perf_counter() - t0
    print(f"native, default threads, runs of {run:3d} poses: {dt / P * 1e3:.3f} ms per pose -> {dt:.3f} s for {P} poses")

from lidar import DualAxisLidarIntrinsics, create_lidar  # noqa: E402
from raycast_engine.raycast_engine_hip import dual_axis_rays_batch  # noqa: E402
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402
kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
poses = poses_from_waypoints(line_trajectory((1.0, 3.0, 1.0), (7.0, 3.0, 1.0), P))
rays = np.empty((P, 64000, 6), dtype=np.float32)
keep = np.ones((P, 64000), dtype=np.uint8)
for rep in range(3):
    np.random.seed(0)
    lidars = [create_lidar(kd, m) for m in poses]
    t0 = time.perf_counter()
    dual_axis_rays_batch(lidars, rays, keep)
    print(f"host ray generation of {P} poses (native draws + numpy trigonometry on the pool): {time.perf_counter() - t0:.3f} s")

#!/usr/bin/env python3
"""tools/c4_host_split.py -- the host side of C4 (BLK2GO, 256 poses) split into its two parts: the seeded draws
(lidarcast.nprandom.scan_draws) and numpy's trigonometry / rotation per pose (IndoorLidar.rays_from_angles), each alone, then
together as raycast_engine_hip.dual_axis_rays_batch runs them."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401
import numpy as np  # noqa: E402
from lidarcast import nprandom  # noqa: E402
from lidar import DualAxisLidarIntrinsics, create_lidar  # noqa: E402
from raycast_engine.raycast_engine_hip import dual_axis_rays_batch, _ray_pool  # noqa: E402
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402

P = 256
kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
poses = poses_from_waypoints(line_trajectory((1.0, 3.0, 1.0), (7.0, 3.0, 1.0), P))
lidars = [create_lidar(kd, m) for m in poses]
n = 64000
rays = np.empty((P, n, 6), dtype=np.float32)
keep = np.ones((P, n), dtype=np.uint8)
np.random.seed(0)
z, u = nprandom.scan_draws(16, 2 * n, n, 0.0, kd.angle_noise_std)


from raycast_engine.raycast_engine_hip import _pose_rays_native as one, _pose_rays_numpy  # noqa: E402

t0 = time.perf_counter()
for i in range(16):
    _pose_rays_numpy(lidars[i], z[i], u[i], rays[i], keep[i])
print(f"angles + rays of one pose, numpy only, one thread: {(time.perf_counter() - t0) / 16 * 1e3:.3f} ms")
th = np.random.default_rng(0).uniform(-1, 1, n)
for name, f in (("cos", np.cos), ("sin", np.sin)):
    t0 = time.perf_counter()
    for _ in range(50):
        f(th)
    print(f"np.{name} of {n} float64: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")

for rep in range(3):
    t0 = time.perf_counter()
    for i in range(16):
        one(lidars[i], z[i], u[i], rays[i], keep[i])
    t1 = time.perf_counter()
    print(f"angles + rays of one pose, one thread: {(t1 - t0) / 16 * 1e3:.3f} ms")
pool = _ray_pool()
for rep in range(3):
    t0 = time.perf_counter()
    fs = [pool.submit(one, lidars[i], z[i % 16], u[i % 16], rays[i], keep[i]) for i in range(P)]
    for f in fs:
        f.result()
    print(f"angles + rays of {P} poses on the pool ({pool._max_workers} threads), draws precomputed: {time.perf_counter() - t0:.4f} s")
for rep in range(3):
    np.random.seed(0)
    t0 = time.perf_counter()
    for a in range(0, P, 16):
        nprandom.scan_draws(16, 2 * n, n, 0.0, kd.angle_noise_std, threads=4)
    print(f"draws alone (4 threads, runs of 16): {time.perf_counter() - t0:.4f} s")
for rep in range(3):
    np.random.seed(0)
    t0 = time.perf_counter()
    dual_axis_rays_batch(lidars, rays, keep)
    print(f"dual_axis_rays_batch: {time.perf_counter() - t0:.4f} s")

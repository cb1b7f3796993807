#!/usr/bin/env python3
"""Per-ray traversal counters of the C3 workload (diagnostic build, LRC_STATS=1): how many inner-node
steps and triangle tests each ray takes, and how well the 64 lanes of a wave agree."""
import os
import sys

os.environ["LRC_STATS"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (sets sys.path for the package)
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidar import IndoorLidar  # noqa: E402

mesh = synth.make_scene(bench.SCENE)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles)
sensor = bench.c3_sensor()
poses = bench.c3_poses(0, 1)[:8]
dirs = IndoorLidar(intrinsics=sensor, pose=np.eye(4)).sensor_directions()
out = scene.scan_poses(poses, dirs, sensor.max_range, want=("t", "normal3"))
st = out["normal3"].reshape(-1, 64, 3)          # waves of 64 consecutive rays
nodes, tris = st[..., 0], st[..., 1]
uni = np.floor(st[..., 2])
dead = np.round((st[..., 2] - uni) * 1024.0)
print("rays", st.shape[0] * 64, "info", scene.info["max_depth"], scene.info["num_nodes"])
for name, a in (("node steps", nodes), ("tri tests", tris)):
    print(f"{name:10s} per ray: mean {a.mean():6.2f}  p50 {np.median(a):5.1f}  p99 {np.percentile(a, 99):6.1f}  max {a.max():5.0f}"
          f"   per wave: mean-of-max {a.max(1).mean():6.2f}  ->  lane efficiency mean/max {a.mean() / a.max(1).mean():.2f}")
print(f"uniform (scalar) node steps per ray: {uni.mean():.2f} = {uni.mean() / nodes.mean():.0%} of node steps")
print(f"dead node steps (no child hit) per ray: {dead.mean():.2f} = {dead.mean() / nodes.mean():.0%} of node steps")
work = nodes * 45 + tris * 40
print(f"work balance inside a wave (mean/max of 45*nodes+40*tris): {work.mean() / work.max(1).mean():.2f}")
# What would K rays per lane with private refill buy (a lane starts its next ray as soon as it finishes one; the wave
# ends when its slowest lane is through)?  Upper bound from the per-ray work: mean/max over lanes of the K-ray sums.
flat_nodes, flat_tris = nodes.reshape(-1), tris.reshape(-1)
w = flat_nodes * 45 + flat_tris * 40
for K in (1, 2, 4, 8):
    for how, arr in (("next tiles of the scanline", w[:len(w) // (64 * K) * 64 * K].reshape(-1, K, 64).sum(1)),):
        print(f"K={K} rays per lane ({how}): work balance mean/max = {arr.mean() / arr.max(1).mean():.2f}")

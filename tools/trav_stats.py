#!/usr/bin/env python3
"""Per-ray traversal counters of the C3 workload (instrumented kernel, lrc_debug_scan_stats): how many inner-node
steps and triangle tests each ray takes, and how well the 64 lanes of a wave agree."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (sets sys.path for the package)
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidar import IndoorLidar  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else bench.SCENE
mesh = synth.make_scene(name)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles)
sensor = bench.c3_sensor()
Lx, Ly, _ = synth.scene_size(name)
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402
poses = poses_from_waypoints(line_trajectory((1.0, Ly / 2, 1.0), (Lx - 1.0, Ly / 2, 1.0), 64))[:8]
dirs = IndoorLidar(intrinsics=sensor, pose=np.eye(4)).sensor_directions()
st = scene.scan_stats(poses, dirs, sensor.max_range).reshape(-1, 64, 5).astype(np.float64)   # waves of 64 consecutive rays
nodes, tris, uni, dead = st[..., 0], st[..., 1], st[..., 2], st[..., 3]
print("scene", name, "pad-clause rejections", int(st[..., 4].sum()))
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
# Lever "waves with few live lanes switch to the four-wide image" (VERDICT r03 item 7a), the counter first: under lock step a
# wave runs max-over-lanes node iterations and lane l is live in the first nodes_l of them, so the iterations run with at most N
# live lanes number c(1) - c(N+1) (c = the wave's per-lane counts, descending).  Their share of ALL node iterations of all waves
# bounds what any change confined to those iterations can save (a four-wide step there halves the iterations at about twice the
# instructions per iteration: it saves latency, not issue).
srt = -np.sort(-nodes, axis=1)
total_iters = srt[:, 0].sum()
for N in (1, 2, 4, 8, 16, 32):
    few = (srt[:, 0] - srt[:, N]).sum()
    print(f"node iterations run with <= {N:2d} live lanes: {few / total_iters:6.1%} of all wave iterations")
longest = np.argsort(-srt[:, 0])[:max(1, len(srt) // 1000)]
print(f"the longest 0.1 % of waves ({len(longest)}): {srt[longest, 0].mean():.0f} iterations, of which with <= 8 live lanes "
      f"{(srt[longest, 0] - srt[longest, 8]).sum() / srt[longest, 0].sum():.1%}, <= 16: {(srt[longest, 0] - srt[longest, 16]).sum() / srt[longest, 0].sum():.1%}")

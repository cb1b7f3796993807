#!/usr/bin/env python3
"""tools/trav_timing.py -- where a wave's time goes (measurement build: build_variants/libtiming.so, -DLRC_EXP_TIMING, whose
statistics kernel stamps s_memtime): total clocks of the traversal, clocks spent waiting for per-lane node fetches.
Run with LRC_LIB=build_variants/libtiming.so on a GPU box; the statistics words are reused: [2] = fetch wait, [3] = total."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidar import IndoorLidar  # noqa: E402
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else bench.SCENE
mesh = synth.make_scene(name)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles)
sensor = bench.c3_sensor()
Lx, Ly, _ = synth.scene_size(name)
poses = poses_from_waypoints(line_trajectory((1.0, Ly / 2, 1.0), (Lx - 1.0, Ly / 2, 1.0), 64))
dirs = IndoorLidar(intrinsics=sensor, pose=np.eye(4)).sensor_directions()
scene.scan_stats(poses, dirs, sensor.max_range)
st = scene.scan_stats(poses, dirs, sensor.max_range).reshape(-1, 64, 5).astype(np.float64)
nodes, tris, wait, total = st[..., 0], st[..., 1], st[..., 2], st[..., 3]
w_total, w_wait = total.max(1), wait.max(1)          # the stamps are wave-level: every lane holds its wave's totals
it_nodes, it_tris = nodes.max(1), tris.max(1)
print(f"scene {name}: {st.shape[0]} waves in one launch")
print(f"traversal clocks per wave: mean {w_total.mean():.0f}  p50 {np.median(w_total):.0f}  p90 {np.percentile(w_total, 90):.0f}")
print(f"of which waiting for per-lane node fetches: mean {w_wait.mean():.0f} = {w_wait.mean() / w_total.mean():.0%}")
print(f"per wave: max node steps over lanes {it_nodes.mean():.1f}, max triangle tests {it_tris.mean():.1f}")
real = st[..., 4].max(1)                               # the same span in 100 MHz ticks (s_memrealtime)
print(f"shader clock during the launch: {w_total.sum() / real.sum() * 0.1:.2f} GHz (s_memtime / s_memrealtime)")

#!/usr/bin/env python3
"""tools/trace_time.py [scene] [lines] [width] [poses] -- HIP-event time of the pose-batched trace kernel for an arbitrary
multi-line sensor (dense_32line elevations resampled to `lines`), for A/B runs of kernel variants (LRC_LIB)."""
import dataclasses
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidar import Indoor8LineLidarIntrinsics, IndoorLidar  # noqa: E402
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402

scene_name = sys.argv[1] if len(sys.argv) > 1 else bench.SCENE
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 32
width = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
P = int(sys.argv[4]) if len(sys.argv) > 4 else 64
mesh = synth.make_scene(scene_name)
Lx, Ly, _ = synth.scene_size(scene_name)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
degs = list(np.linspace(15.0, -20.0, lines))
k = Indoor8LineLidarIntrinsics(vertical_res=lines, horizontal_res=width, max_range=25.0, vertical_degrees=degs)
dirs = IndoorLidar(k, np.eye(4)).sensor_directions()
poses = poses_from_waypoints(line_trajectory((1.0, Ly / 2, 1.0), (Lx - 1.0, Ly / 2, 1.0), P))
dev = torch.device("cuda", 0)
n = P * len(dirs)
want = tuple(os.environ.get("LRC_TT_WANT", "t,prim,normal3,point3,sem,ins,tile_count").split(","))
hits = lidarcast.DeviceHits(n, dev, want=want)
d_poses, d_dirs = torch.from_numpy(poses.reshape(P, 16)).to(dev), torch.from_numpy(dirs).to(dev)
st = torch.cuda.current_stream().cuda_stream
grid = None
if os.environ.get("LRC_GRID", "0") != "0":      # packet kernel (lrc_scan_grid_dev)
    from raycast_engine.raycast_engine_hip import RaycastEngineHIP
    grid = RaycastEngineHIP._derive_grid(dirs, width)
    assert grid is not None, "the table is not a grid the packet kernel takes"
for _ in range(5):
    scene.scan_poses_dev(d_poses, d_dirs, hits, k.max_range, st, grid=grid)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
for a, b in ev:
    a.record()
    scene.scan_poses_dev(d_poses, d_dirs, hits, k.max_range, st, grid=grid)
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
chk = int(hits["prim"].to(torch.int64).sum().item())
print(f"{'packet ' if grid else ''}{scene_name} {lines}x{width} x{P}: {n} rays, median {ms[len(ms)//2]:.4f} ms, min {ms[0]:.4f} ms, "
      f"{n / ms[len(ms)//2] / 1e6:.2f} G rays/s, checksum {chk}")

#!/usr/bin/env python3
"""Print a SHA-256 over the hit records of a fixed small scan.  Run under different LRC_* kernel variants
(LRC_UNIFORM, LRC_LEAFW, LRC_SPEC, LRC_MAX_LEAF): the digest must not change (tests/test_parity_gpu.py)."""
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402
from lidar import Indoor8LineLidarIntrinsics, IndoorLidar  # noqa: E402
from lidarcast import synth  # noqa: E402

mesh = synth.make_room(size=(4.0, 3.0, 2.5), num_boxes=4, seed=5, cell=0.04)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
# 16 lines x 1024: 256 tiles per pose, so the default launch deals every pose to the XCDs in 16 chunks (XCD striping)
k = Indoor8LineLidarIntrinsics(vertical_res=16, horizontal_res=1024, max_range=20.0,
                               vertical_degrees=[float(x) for x in np.linspace(25.0, -35.0, 16)])
poses = np.stack([np.eye(4) for _ in range(4)])
poses[:, :3, 3] = [(0.8, 1.2, 1.0), (1.6, 1.4, 1.0), (2.4, 1.6, 1.1), (3.2, 1.5, 0.9)]
dirs = IndoorLidar(k, np.eye(4)).sensor_directions()
out = scene.scan_poses(poses, dirs, k.max_range)
# the same scan through the grid entry point (lrc_scan_grid_compact): the per-ray kernel in the product library, the packet
# kernel in the laboratory build (LRC_SECTOR, default on there)
from raycast_engine.raycast_engine_hip import RaycastEngineHIP  # noqa: E402
grid = RaycastEngineHIP._derive_grid(dirs, k.horizontal_res)
assert grid is not None
frames = scene.scan_poses_compact(poses, dirs, k.max_range, want=("point3", "sem", "ins", "incident_deg"), grid=grid)
rng = np.random.default_rng(1)
soup = rng.uniform(-3, 3, (3000, 1, 3)) + rng.normal(scale=0.4, size=(3000, 3, 3))
scene2 = lidarcast.Scene(ctx, soup.reshape(-1, 3), np.arange(9000).reshape(-1, 3))
o = rng.uniform(-3, 3, (20000, 3))
d = rng.normal(size=(20000, 3))
out2 = scene2.cast(np.concatenate([o, d], 1).astype(np.float32))
# a scan of more tiles than the chip holds waves at once (16 x 1024 x 40 poses = 10 240 tiles against 8 192 wave slots)
many = np.stack([np.eye(4) for _ in range(40)])
many[:, :3, 3] = np.stack([np.linspace(0.6, 3.4, 40), np.linspace(1.0, 2.0, 40), np.full(40, 1.0)], 1)
out3 = scene.scan_poses(many, dirs, k.max_range, want=("t", "prim", "normal3", "point3", "sem", "ins"))
h = hashlib.sha256()
for key in ("t", "prim", "normal3", "point3", "sem", "ins"):
    h.update(out3[key].tobytes())
for res in (out, out2):
    for key in ("t", "prim", "normal3", "point3", "sem", "ins", "incident_deg"):
        h.update(res[key].tobytes())
for key in ("point3", "sem", "ins", "incident_deg", "counts"):
    h.update(np.ascontiguousarray(frames[key]).tobytes())
assert frames["total"] == int(np.isfinite(out["t"]).sum())
print(h.hexdigest(), int(np.isfinite(out["t"]).sum()), int(np.isfinite(out2["t"]).sum()))

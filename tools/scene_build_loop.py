#!/usr/bin/env python3
"""tools/scene_build_loop.py [scene] [repeats] -- build the same scene repeatedly (lrc_scene_create, mesh in host memory):
the command whose rocprofv3 --kernel-trace --stats summary prices the kernels of the device builder."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else bench.SCENE
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mesh = synth.make_scene(name)
v = np.ascontiguousarray(mesh.vertices, dtype=np.float32)
f = np.ascontiguousarray(mesh.triangles, dtype=np.uint32)
ctx = lidarcast.Context(0)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    sc = lidarcast.Scene(ctx, v, f, mesh.triangle_sem, mesh.triangle_ins)
    ts.append((time.perf_counter() - t0) * 1e3)
    info = sc.info
    sc.close()
print(f"{name}: T={info['num_triangles']} nodes={info['num_nodes']} depth={info['max_depth']} "
      f"create ms: first {ts[0]:.2f}, median {np.median(ts[1:]):.2f} (transfer {info['upload_ms']:.2f} + build {info['build_ms']:.2f})")

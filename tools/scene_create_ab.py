#!/usr/bin/env python3
"""tools/scene_create_ab.py -- lrc_scene_create on the C3 scene, 40 creates after 5 warm ones: median / min wall time and the
library's own split (upload, hierarchy, emit).  Run once per library (LRC_LIB) on the same box for an A/B of the builder."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402

mesh = synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else bench.SCENE)
ctx = lidarcast.Context(0)
ts, infos = [], []
for i in range(45):
    t0 = time.perf_counter()
    sc = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
    dt = time.perf_counter() - t0
    if i >= 5:
        ts.append(dt * 1e3)
        infos.append(sc.info)
    sc.close()
ts = np.array(ts)
print(f"{os.environ.get('LRC_LIB', 'in-tree'):40s} create median {np.median(ts):.3f} ms  min {ts.min():.3f}  "
      f"build {np.median([i['build_ms'] for i in infos]):.3f}  upload {np.median([i['upload_ms'] for i in infos]):.3f}")

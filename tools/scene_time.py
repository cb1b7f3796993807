"""tools/scene_time.py -- wall time of lrc_scene_create on the C3 scene: host BVH build vs upload (first call pays HIP start-up)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import numpy as np
import lidarcast
from lidarcast import synth
mesh = synth.make_scene(bench.SCENE)
ctx = lidarcast.Context(0)
for i in range(3):
    t0 = time.perf_counter()
    sc = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
    dt = time.perf_counter() - t0
    inf = sc.info
    print(f"create {dt*1e3:.1f} ms  build {inf['build_ms']:.1f}  upload {inf['upload_ms']:.1f}  bytes {inf['device_bytes']/1e6:.1f} MB")
    sc.close()

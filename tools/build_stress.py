#!/usr/bin/env python3
"""tools/build_stress.py [scenes] [seed] -- the device scene build against the host builder over many random scenes (the
geometry generator of tools/qnode_stress.py: a centimetre to 300 m, flat and degenerate extents, 1 .. 20 000 triangles), random
leaf size / depth slack / layout head: every array the trace kernels read must be byte-identical; prints the mismatches."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
ctx = lidarcast.Context(0)
ARRAYS = ("nodes", "tris", "slot_prim", "slot_label", "prim_plane", "nodes_q", "nodes_n")
bad = tri_total = dev_built = 0
for case in range(N):
    scale = 10.0 ** rng.uniform(-2.0, 2.5)
    T = int(rng.integers(1, 20000))
    centre = rng.uniform(-1.0, 1.0, 3) * scale * rng.choice([0.0, 0.5, 2.0])
    ext = np.array([1.0, rng.uniform(0.05, 1.0), rng.uniform(0.0 if case % 17 == 3 else 0.01, 1.0)])[rng.permutation(3)] * scale
    c = rng.uniform(-0.5, 0.5, (T, 1, 3)) * ext
    if case % 5 == 0:                      # many coincident centroids: equal costs, median fallbacks
        c = np.round(c / (0.1 * scale)) * (0.1 * scale)
    tri = (centre + c + rng.normal(scale=10.0 ** rng.uniform(-3, -1) * scale, size=(T, 3, 3)) * (ext > 0)).astype(np.float32)
    v, f = tri.reshape(-1, 3), np.arange(3 * T, dtype=np.int32).reshape(-1, 3)
    sem = rng.integers(0, 13, T).astype(np.uint16)
    ins = rng.integers(0, 500, T).astype(np.uint16)
    env = {"LRC_MAX_LEAF": str(int(rng.integers(1, 5))), "LRC_DEPTH_SLACK": str(int(rng.integers(0, 4))),
           "LRC_BFS_NODES": str(int(rng.choice([1, 7, 256, 4096, 10 ** 6])))}
    os.environ.update(env)
    scenes = []
    for dev in ("0", "1"):
        os.environ["LRC_DEVICE_BUILD"] = dev
        scenes.append(lidarcast.Scene(ctx, v, f, sem, ins))
    host, devs = scenes
    dev_built += int(devs.info.get("device_build", 0))
    diff = [a for a in ARRAYS if not np.array_equal(host.export_array(a), devs.export_array(a))]
    if diff:
        bad += 1
        print(f"case {case}: T={T} scale={scale:.3g} {env}: arrays differ: {diff}")
    tri_total += T
    host.close(); devs.close()
print(f"{N} scenes ({dev_built} built on the device), {tri_total} triangles: {bad} scenes differ")

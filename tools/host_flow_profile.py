#!/usr/bin/env python3
"""tools/host_flow_profile.py -- cProfile of the reference's per-scene flow on the C3 workload (load_scene -> run_simulation ->
scene statistics -> the assembled scene cloud): where the HOST time of a caller goes besides the scan itself."""
import cProfile
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
from lidarcast import synth  # noqa: E402
from s3dis_simulator import S3DISSimulator  # noqa: E402
from trajectory import Waypoint  # noqa: E402

mesh = synth.make_scene(bench.SCENE)
poses = bench.c3_poses(0, 1)
wps = [Waypoint(m[0, 3], m[1, 3], m[2, 3], yaw=0.0, timestamp=float(i)) for i, m in enumerate(poses)]
sim = S3DISSimulator({"raycast_engine": {"use_gpu": True}})
sim.lidar_config = bench.c3_sensor()


def flow():
    sim.load_scene(mesh, "bench")
    sc = sim.run_simulation(wps)
    st = sc.compute_statistics() if hasattr(sc, "compute_statistics") else None
    pts = sc.combined_points()
    lab = sc.combined_labels()
    return sc, st, pts, lab


for _ in range(3):
    flow()
ts = []
for _ in range(7):
    t0 = time.perf_counter()
    flow()
    ts.append((time.perf_counter() - t0) * 1e3)
print("flow ms: median %.2f min %.2f" % (np.median(ts), min(ts)))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    flow()
pr.disable()
pstats.Stats(pr).strip_dirs().sort_stats("tottime").print_stats(25)

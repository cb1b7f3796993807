#!/usr/bin/env python3
"""tools/run_sim_profile.py -- where S3DISSimulator.run_simulation spends its time on the C3 workload (cProfile)."""
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
sim = S3DISSimulator({"raycast_engine": {"use_gpu": True, "eager_labels": os.environ.get("EAGER_LABELS", "0") == "1"}})
sim.lidar_config = bench.c3_sensor()
sim.load_scene(mesh, "bench")
poses = bench.c3_poses(0, 1)
wps = [Waypoint(m[0, 3], m[1, 3], m[2, 3], yaw=0.0, timestamp=float(i)) for i, m in enumerate(poses)]
for _ in range(3):
    t0 = time.perf_counter()
    sc = sim.run_simulation(wps)
    print("run_simulation ms", (time.perf_counter() - t0) * 1e3)
    del sc
pr = cProfile.Profile()
pr.enable()
sc = sim.run_simulation(wps)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
# the per-waypoint drop-in path of the unmodified reference loop
from lidar import create_lidar  # noqa: E402
eng = sim.raycast_engine
for _ in range(2):
    t0 = time.perf_counter()
    keepalive = [eng.lidar_intersect_mesh(create_lidar(sim.lidar_config, m), mesh) for m in poses]
    dt = time.perf_counter() - t0
print(f"per-waypoint lidar_intersect_mesh x {len(poses)}: {dt * 1e3:.2f} ms total, {dt / len(poses) * 1e3:.3f} ms per pose, "
      f"{len(poses) * 65536 / dt / 1e6:.1f} M rays/s")

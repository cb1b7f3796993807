#!/usr/bin/env python3
"""tools/c5_profile.py -- where the time of a C5 batch (six scenes x 64 poses through run_scene_batch) goes on the host."""
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
from s3dis_simulator import run_scene_batch  # noqa: E402
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402

sensor = bench.c3_sensor()
names = list(synth.SCENES)
meshes = {n: synth.make_scene(n) for n in names}


def poses(n):
    Lx, Ly, _ = synth.scene_size(n)
    return poses_from_waypoints(line_trajectory((1.0, Ly / 2, 1.0), (Lx - 1.0, Ly / 2, 1.0), 64))


traj = {n: poses(n) for n in names}
seen = []
for rep in range(3):
    t0 = time.perf_counter()
    r = run_scene_batch([(n, meshes[n]) for n in names], traj, sensor=sensor, config={"raycast_engine": {"use_gpu": True}},
                        on_scene=lambda name, sc: seen.append(len(sc.frames)))
    print(f"batch {rep}: wall {time.perf_counter() - t0:.4f} s; scan stages {r['seconds']:.4f} s, builds {r['build_seconds']:.4f} s; "
          + " ".join(f"{n}:{v['seconds'] * 1e3:.1f}+{v['build_seconds'] * 1e3:.1f}" for n, v in r["scenes"].items()))
pr = cProfile.Profile()
pr.enable()
run_scene_batch([(n, meshes[n]) for n in names], traj, sensor=sensor, config={"raycast_engine": {"use_gpu": True}},
                on_scene=lambda name, sc: seen.append(len(sc.frames)))
pr.disable()
pstats.Stats(pr).strip_dirs().sort_stats("cumulative").print_stats(28)

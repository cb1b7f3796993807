#!/usr/bin/env python3
"""tools/per_waypoint_time.py -- where one per-waypoint drop-in call (RaycastEngineGPU.lidar_intersect_mesh, 65 536 rays)
spends its time: scene cache lookup, direction table, the library call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, numpy as np
from lidarcast import synth
from lidar import create_lidar
from raycast_engine import RaycastEngineGPU
mesh = synth.make_scene(bench.SCENE)
eng = RaycastEngineGPU()
k = bench.c3_sensor(); poses = bench.c3_poses(0,1)
l = create_lidar(k, poses[5])
eng.lidar_intersect_mesh(l, mesh)
def t(f, n=30):
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); f(); ts.append(time.perf_counter()-t0)
    return np.median(ts)*1e3
print("lidar_intersect_mesh ms", t(lambda: eng.lidar_intersect_mesh(l, mesh)))
print("scene_for ms", t(lambda: eng.scene_for(mesh)))
sc = eng.scene_for(mesh); tab = eng._direction_table(k)
print("direction_table ms", t(lambda: eng._direction_table(k)))
p1 = np.asarray(l.pose)[None]
print("scan_poses_compact(P=1) ms", t(lambda: sc.scan_poses_compact(p1, tab, k.max_range, want=("point3","incident_deg"))))
print("scan_poses_compact(P=1, point3 only) ms", t(lambda: sc.scan_poses_compact(p1, tab, k.max_range, want=("point3",))))
print("scan_poses records (P=1) ms", t(lambda: sc.scan_poses(p1, tab, k.max_range, want=("t","point3","incident_deg"))))

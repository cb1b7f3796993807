#!/usr/bin/env python3
"""tools/compact_time.py -- wall time of the frame-producing call (lrc_scan_poses_compact) on the C3 workload for several
sets of requested columns, with and without the device-side per-pose statistics."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, numpy as np, lidarcast
from lidarcast import synth
from lidar import IndoorLidar
mesh = synth.make_scene(bench.SCENE)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
sensor = bench.c3_sensor(); poses = bench.c3_poses(0,1)
dirs = IndoorLidar(sensor, np.eye(4)).sensor_directions()
for want in (("point3","sem","ins"), ("point3","sem","ins","range_origin"), ("point3","sem","ins","range_origin_stats"), ("point3",), ("point3","sem","ins","incident_deg","incident_stats","range_origin_stats")):
    ts=[]
    for _ in range(8):
        t0=time.perf_counter(); fr = scene.scan_poses_compact(poses, dirs, sensor.max_range, want=want); ts.append(time.perf_counter()-t0); del fr
    print(want, "median ms %.3f min %.3f" % (np.median(ts[2:])*1e3, min(ts)*1e3))

import os, sys, time, cProfile, pstats
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO)
import bench
import numpy as np
from lidarcast import synth
from s3dis_simulator import S3DISSimulator
from trajectory import Waypoint
mesh = synth.make_scene(bench.SCENE)
sim = S3DISSimulator({"raycast_engine": {"use_gpu": True}})
sim.lidar_config = bench.c3_sensor()
sim.load_scene(mesh, "bench")
poses = bench.c3_poses(0, 1)
wps = [Waypoint(m[0, 3], m[1, 3], m[2, 3], yaw=0.0, timestamp=float(i)) for i, m in enumerate(poses)]
for _ in range(4):
    sc = sim.run_simulation(wps); del sc
ts = []
for _ in range(15):
    t0 = time.perf_counter(); sc = sim.run_simulation(wps); ts.append((time.perf_counter() - t0) * 1e3); del sc
print("run_simulation ms: median %.3f min %.3f" % (np.median(ts), min(ts)))
eng = sim.raycast_engine
from trajectory import poses_from_waypoints
P = poses_from_waypoints(wps)
ts = []
want = ("point3", "range_origin_stats")
for _ in range(15):
    t0 = time.perf_counter(); fr = eng.scan_frames(sim.lidar_config, P, mesh, want=want); ts.append((time.perf_counter() - t0) * 1e3); del fr
print("scan_frames(point3 + range stats) ms: median %.3f min %.3f" % (np.median(ts), min(ts)))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    sc = sim.run_simulation(wps); del sc
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

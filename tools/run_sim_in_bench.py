#!/usr/bin/env python3
"""tools/run_sim_in_bench.py -- why S3DISSimulator.run_simulation takes 2.0-2.2 ms inside bench.py and 1.4 ms alone: the same
15 timed calls after each thing bench.py has done by then (torch's HIP context, the bench's own scene + record buffers + scan
pipeline, a few thousand device steps), and with the cyclic garbage collector off."""
import gc
import os
import sys
import time

REPO = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidar import IndoorLidar  # noqa: E402
from s3dis_simulator import S3DISSimulator  # noqa: E402
from trajectory import Waypoint  # noqa: E402

mesh = synth.make_scene(bench.SCENE)
sensor = bench.c3_sensor()
poses = bench.c3_poses(0, 1)
wps = [Waypoint(m[0, 3], m[1, 3], m[2, 3], yaw=0.0, timestamp=float(i)) for i, m in enumerate(poses)]


def fresh_sim():
    sim = S3DISSimulator({"raycast_engine": {"use_gpu": True}})
    sim.lidar_config = sensor
    sim.load_scene(mesh, "bench")
    return sim


def timed(what, sim, reps=15):
    for _ in range(3):
        sc = sim.run_simulation(wps)
        del sc
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        sc = sim.run_simulation(wps)
        ts.append((time.perf_counter() - t0) * 1e3)
        del sc
    print(f"{what:70s} median {np.median(ts):.3f} ms  min {min(ts):.3f}  max {max(ts):.3f}", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "alone":
    timed("alone (no torch in the process)", fresh_sim())
    sys.exit(0)
import torch  # noqa: E402  (torch must initialise HIP before this library does, as in bench.py)
dev = torch.device("cuda", 0)
x = torch.zeros(1, device=dev)
torch.cuda.synchronize()
sim = fresh_sim()
timed("after import torch + its HIP context", sim)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
dirs = IndoorLidar(sensor, np.eye(4)).sensor_directions()
P, N = len(poses), len(dirs)
hits = lidarcast.DeviceHits(P * N, dev, want=("t", "prim", "normal3", "point3", "sem", "ins", "tile_count"))
rows = [torch.zeros((P * N, 4), dtype=torch.float32, device=dev) for _ in range(3)]
counts = [torch.zeros(P, dtype=torch.int64, device=dev) for _ in range(3)]
d_poses, d_dirs = torch.from_numpy(poses.reshape(P, 16)).to(dev), torch.from_numpy(dirs).to(dev)
pipe = lidarcast.ScanPipe(scene, P, N)
st = torch.cuda.current_stream().cuda_stream
timed("after the bench's own scene, record buffers and scan pipeline", sim)
for i in range(3000):
    pipe.submit(d_poses, d_dirs, sensor.max_range, out_rows_t=rows[i % 3], counts_t=counts[i % 3], stream=st)
pipe.wait(st)
torch.cuda.synchronize()
timed("after 3000 pipelined device steps", sim)
sim2 = fresh_sim()
timed("a second simulator (as bench.py creates its own)", sim2)
gc.disable()
timed("the same with gc.disable()", sim2)
gc.enable()
gc.freeze()
timed("the same with gc.freeze() (collector on, start-up objects exempt)", sim2)

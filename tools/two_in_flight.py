#!/usr/bin/env python3
"""tools/two_in_flight.py [scene] -- what a caller gains by keeping two trace launches in flight (two streams, two record
sets): the tail of one launch is filled by the head of the next ("Where a 64-pose launch loses its
time").  C3 sensor x 64 poses per launch; trace kernel alone, no compaction."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidar import IndoorLidar  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else bench.SCENE
mesh = synth.make_scene(name)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
sensor = bench.c3_sensor()
poses = bench.c3_poses(0, 1)
P = len(poses)
dirs = IndoorLidar(sensor, np.eye(4)).sensor_directions()
dev = torch.device("cuda", 0)
n = P * len(dirs)
want = ("t", "prim", "normal3", "point3", "sem", "ins", "tile_count")
sets = [lidarcast.DeviceHits(n, dev, want=want) for _ in range(2)]
d_poses, d_dirs = torch.from_numpy(poses.reshape(P, 16)).to(dev), torch.from_numpy(dirs).to(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
K = 200
ctx.set_launch_chaining(os.environ.get("PIPE_CHAIN", "0") == "1")       # opt-in (off by default)
print("launch chaining (enabled, supported):", ctx.launch_chaining(), flush=True)


def run(nstreams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        s = streams[i % nstreams]
        scene.scan_poses_dev(d_poses, d_dirs, sets[i % nstreams], sensor.max_range, s.cuda_stream)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for rep in range(3):
    a, b = run(1), run(2)
    print(f"{name}: {K} launches of {n} rays: one stream {a / K * 1e3:.4f} ms per launch = {n * K / a / 1e9:.2f} G rays/s; "
          f"two streams {b / K * 1e3:.4f} ms = {n * K / b / 1e9:.2f} G rays/s ({(a / b - 1) * 100:+.1f} %)")

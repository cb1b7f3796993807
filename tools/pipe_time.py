#!/usr/bin/env python3
"""tools/pipe_time.py [scene] -- the C3 step through lrc_pipe_* against the same step as two calls on one stream
(lrc_scan_poses_dev + lrc_compact_dev), same box, alternating; and that both produce the same bytes."""
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
from lidarcast._capi import LrcCompactIO  # noqa: E402
from lidar import IndoorLidar  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else bench.SCENE
K = int(os.environ.get("PIPE_STEPS", "300"))
mesh = synth.make_scene(name)
ctx = lidarcast.Context(0)
ctx.set_launch_chaining(os.environ.get("PIPE_CHAIN", "0") == "1")       # opt-in (off by default)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
sensor = bench.c3_sensor()
poses = bench.c3_poses(0, 1)
P = len(poses)
dirs = IndoorLidar(sensor, np.eye(4)).sensor_directions()
N = len(dirs)
dev = torch.device("cuda", 0)
n = P * N
want = ("t", "prim", "normal3", "point3", "sem", "ins", "tile_count")
hits = lidarcast.DeviceHits(n, dev, want=want)
clouds = [torch.zeros((n, 4), dtype=torch.float32, device=dev) for _ in range(4)]
counts = [torch.zeros(P, dtype=torch.int64, device=dev) for _ in range(4)]
d_poses, d_dirs = torch.from_numpy(poses.reshape(P, 16)).to(dev), torch.from_numpy(dirs).to(dev)
stream = torch.cuda.current_stream().cuda_stream
io = LrcCompactIO()
io.t, io.point3, io.sem, io.ins = hits["t"].data_ptr(), hits["point3"].data_ptr(), hits["sem"].data_ptr(), hits["ins"].data_ptr()
io.tile_count = hits["tile_count"].data_ptr()
io.counts, io.out_xyzl = counts[3].data_ptr(), clouds[3].data_ptr()
pipe = lidarcast.ScanPipe(scene, P, N)
print("launch chaining (enabled, supported):", ctx.launch_chaining(), flush=True)


def serial():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        scene.scan_poses_dev(d_poses, d_dirs, hits, sensor.max_range, stream)
        ctx.compact_dev(P, N, io, stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


def piped():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        last = pipe.submit(d_poses, d_dirs, sensor.max_range, out_rows_t=clouds[i % 3], counts_t=counts[i % 3], stream=stream)
    pipe.wait(stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K, last


for rep in range(4):
    a = serial()
    b, last = piped()
    print(f"{name}: serial {a * 1e3:.4f} ms/step = {n / a / 1e9:.2f} G rays/s | pipeline {b * 1e3:.4f} ms/step = "
          f"{n / b / 1e9:.2f} G rays/s ({(a / b - 1) * 100:+.1f} %) | in-pipeline trace launch {pipe.trace_ms(last):.4f} ms", flush=True)
k = int(counts[3].sum().item())
for j in range(3):
    assert torch.equal(counts[j], counts[3]), "per-pose counts differ"
    assert torch.equal(clouds[j][:k].view(torch.int32), clouds[3][:k].view(torch.int32)), "rows differ"
print(f"pipeline rows == serial rows ({k} rows, bit for bit)")

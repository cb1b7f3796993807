#!/usr/bin/env python3
"""Row N1 measurement: 1-NN label lookup of the C3 hit cloud (4.19 M points) in a 1 M-point annotated cloud.
GPU: lrc_nn_query_dev timed with events on the stream; CPU: the reference's own call
(sklearn NearestNeighbors ball_tree, containers/s3dis_sim_scene.py:416-418) on a bounded sample."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lidarcast  # noqa: E402
from lidar import IndoorLidar  # noqa: E402
from lidarcast import synth  # noqa: E402

mesh = synth.make_scene(bench.SCENE)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles)
sensor = bench.c3_sensor()
poses = bench.c3_poses(0, 1)
dirs = IndoorLidar(sensor, np.eye(4)).sensor_directions()
rec = scene.scan_poses(poses, dirs, sensor.max_range, want=("t", "point3"))
cloud = rec["point3"][np.isfinite(rec["t"])]

rng = np.random.default_rng(0)
M = 1_000_000
tri = mesh.triangles[rng.integers(0, len(mesh.triangles), M)]
w = rng.dirichlet([1, 1, 1], M)
ann = (mesh.vertices[tri] * w[:, :, None]).sum(1) + rng.normal(0, 0.003, (M, 3))

t0 = time.perf_counter()
nn = lidarcast.NearestIndex(ctx, ann)
build_s = time.perf_counter() - t0
dev = torch.device("cuda", 0)
q = torch.from_numpy(cloud).to(dev)
idx = torch.empty(len(cloud), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    nn.query_dev(q, idx, None, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    nn.query_dev(q, idx, None, st)
e1.record()
torch.cuda.synchronize()
gpu_ms = e0.elapsed_time(e1) / 10

from sklearn.neighbors import NearestNeighbors  # noqa: E402
t0 = time.perf_counter()
tree = NearestNeighbors(n_neighbors=1, algorithm="ball_tree").fit(ann)
fit_s = time.perf_counter() - t0
sample = cloud[:: max(1, len(cloud) // 200_000)][:200_000]
t0 = time.perf_counter()
ref = tree.kneighbors(sample)[1][:, 0]
cpu_s = time.perf_counter() - t0
got = idx.cpu().numpy().view(np.uint32)[:: max(1, len(cloud) // 200_000)][:200_000]
print(json.dumps({
    "annotated_points": M, "queries": int(len(cloud)), "gpu_query_ms": gpu_ms,
    "gpu_queries_per_s": len(cloud) / (gpu_ms * 1e-3), "gpu_index_build_s": build_s,
    "cpu_sklearn_ball_tree_queries_per_s": len(sample) / cpu_s, "cpu_sample": int(len(sample)),
    "cpu_fit_s": fit_s, "indices_identical_on_sample": bool(np.array_equal(got, ref.astype(np.uint32))),
}, indent=1))

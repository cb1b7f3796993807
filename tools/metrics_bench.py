#!/usr/bin/env python3
"""Row N3 measurement: the reference's sampled metrics on two 1 M-point clouds, GPU (host-array API, copies
included) vs the dense-numpy formulation the reference uses (evaluate_single_scene.py:55-111)."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401
import numpy as np  # noqa: E402
from lidarcast import metrics  # noqa: E402

rng = np.random.default_rng(0)
X = rng.uniform(0, 5, (1_000_000, 3)).astype(np.float32)
Y = (X + rng.normal(0, 0.01, X.shape)).astype(np.float32)


def numpy_mmd(X, Y, max_points=10000, gamma=1.0):
    Xs, Ys = metrics.sample_points(X, max_points), metrics.sample_points(Y, max_points)

    def k(A, B):
        d = np.sum(A ** 2, 1)[:, None] + np.sum(B ** 2, 1)[None, :] - 2 * np.dot(A, B.T)
        return np.exp(-gamma * np.maximum(d, 0))
    m, n = len(Xs), len(Ys)
    return np.sum(k(Xs, Xs)) / (m * m) + np.sum(k(Ys, Ys)) / (n * n) - 2 * np.sum(k(Xs, Ys)) / (m * n)


def numpy_cd(X, Y):
    Xs, Ys = metrics.sample_points(X, 5000), metrics.sample_points(Y, 5000)
    d = np.linalg.norm(Xs[:, None] - Ys, axis=2)
    return np.mean(d.min(1)) + np.mean(d.min(0))


out = {}
metrics.compute_mmd_sampled(X[:1000], Y[:1000])          # warm-up (context, kernels)
for name, gpu, cpu in (("mmd_10000", lambda: metrics.compute_mmd_sampled(X, Y), lambda: numpy_mmd(X, Y)),
                       ("chamfer_5000", lambda: metrics.compute_chamfer_distance(X, Y), lambda: numpy_cd(X, Y))):
    np.random.seed(1); t0 = time.perf_counter(); g = gpu(); tg = time.perf_counter() - t0
    np.random.seed(1); t0 = time.perf_counter(); c = cpu(); tc = time.perf_counter() - t0
    out[name] = {"gpu_s": tg, "numpy_s": tc, "gpu_value": float(g), "numpy_value": float(c)}
print(json.dumps(out, indent=1))

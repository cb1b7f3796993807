#!/usr/bin/env python3
"""Capture golden values of the REFERENCE's cloud metrics (run in the build container only).

    python tests/golden/make_metrics_golden.py        # writes tests/golden/metrics_golden.npz

Imports /root/reference/evaluate_single_scene.py (never copied).  The module imports ``open3d`` at the top only
for its PLY loader, which is not called here; an empty stand-in module is registered so the import succeeds.
The metric functions themselves are pure numpy and run verbatim: compute_mmd_sampled (:55-79),
compute_chamfer_distance (:81-96), compute_hausdorff_distance (:98-111), with np.random.seed fixed before each
call so that the sampled subsets are reproducible."""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
import numpy as np  # noqa: E402

sys.modules.setdefault("open3d", types.ModuleType("open3d"))
spec = importlib.util.spec_from_file_location("ref_evaluate_single_scene", os.path.join(REF, "evaluate_single_scene.py"))
ev = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ev)


def surface_cloud(n, seed, size, jitter):
    rng = np.random.default_rng(seed)
    p = rng.uniform([0, 0, 0], size, size=(n, 3))
    face = rng.integers(0, 6, n)
    for a in range(3):
        p[face == 2 * a, a] = 0.0
        p[face == 2 * a + 1, a] = size[a]
    return (p + rng.normal(0, jitter, p.shape)).astype(np.float32)


X = surface_cloud(14000, 1, (5.0, 4.0, 3.0), 0.002)
Y = surface_cloud(12000, 2, (5.2, 3.9, 3.0), 0.01)
out = {"X": X, "Y": Y}
for tag, (a, b) in {"xy": (X, Y), "xx": (X, X), "small": (X[:2000], Y[:1500])}.items():
    np.random.seed(123)
    out[f"{tag}_cd"] = np.float64(ev.compute_chamfer_distance(a, b))
    np.random.seed(124)
    out[f"{tag}_hd"] = np.float64(ev.compute_hausdorff_distance(a, b))
    np.random.seed(125)
    out[f"{tag}_mmd"] = np.float64(ev.compute_mmd_sampled(a, b, max_points=4000, gamma=1.0))
    print(tag, out[f"{tag}_cd"], out[f"{tag}_hd"], out[f"{tag}_mmd"])
sx, sy = ev.analyze_point_cloud(X, "x"), ev.analyze_point_cloud(Y, "y")
out["X_volume"], out["X_density"] = np.float64(sx["volume"]), np.float64(sx["density"])
out["vol_compat"] = np.array(ev.check_volume_compatibility(sx["volume"], sy["volume"], 0.3), dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "metrics_golden.npz"), **out)
print("wrote", os.path.getsize(os.path.join(HERE, "metrics_golden.npz")), "bytes")

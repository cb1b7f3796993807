"""Sampled point-cloud similarity metrics of the validation stage (reference: evaluate_single_scene.py:47-133).

Same function names, sampling (``np.random.choice(len, k, replace=False)`` on the global stream, so a seeded run
draws the same subsets as the reference) and definitions; the dense O(n*m) distance / kernel matrices the
reference builds in numpy are evaluated by HIP kernels (csrc/lrc_metrics.hip)."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check

_ctx = None


def _context():
    global _ctx
    if _ctx is None:
        from .core import Context
        _ctx = Context(0)
    return _ctx


def _f32(a):
    a = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("point clouds must be (N, 3)")
    return a


def sample_points(points, max_points=10000):
    if len(points) <= max_points:
        return points
    return points[np.random.choice(len(points), max_points, replace=False)]


def min_distances(A, B, ctx=None):
    """out[i] = min_j |A_i - B_j| (float32)."""
    A, B = _f32(A), _f32(B)
    out = np.empty(len(A), dtype=np.float32)
    ctx = ctx or _context()
    check(_capi.load().lrc_min_distances(ctx._h, A.ctypes.data, len(A), B.ctypes.data, len(B), out.ctypes.data),
          "lrc_min_distances")
    return out


def rbf_kernel_sum(A, B, gamma=1.0, ctx=None):
    A, B = _f32(A), _f32(B)
    s = C.c_double(0.0)
    ctx = ctx or _context()
    check(_capi.load().lrc_rbf_kernel_sum(ctx._h, A.ctypes.data, len(A), B.ctypes.data, len(B), float(gamma),
                                          C.byref(s)), "lrc_rbf_kernel_sum")
    return s.value


def compute_chamfer_distance(X, Y):
    Xs, Ys = sample_points(X, 5000), sample_points(Y, 5000)
    return float(np.mean(min_distances(Xs, Ys)) + np.mean(min_distances(Ys, Xs)))


def compute_hausdorff_distance(X, Y):
    Xs, Ys = sample_points(X, 3000), sample_points(Y, 3000)
    return float(max(np.max(min_distances(Xs, Ys)), np.max(min_distances(Ys, Xs))))


def compute_mmd_sampled(X, Y, max_points=10000, gamma=1.0):
    Xs, Ys = sample_points(X, max_points), sample_points(Y, max_points)
    m, n = len(Xs), len(Ys)
    return (rbf_kernel_sum(Xs, Xs, gamma) / (m * m) + rbf_kernel_sum(Ys, Ys, gamma) / (n * n)
            - 2 * rbf_kernel_sum(Xs, Ys, gamma) / (m * n))


def normalize_coordinates(points, method="center"):
    if method in ("center", "zero_center"):
        return points - (points.min(axis=0) + points.max(axis=0)) / 2
    if method == "min":
        return points - points.min(axis=0)
    return points


def analyze_point_cloud(points, name="", normalize=True):
    p = normalize_coordinates(points, "zero_center") if normalize else points
    ext = p.max(axis=0) - p.min(axis=0)
    volume = float(ext[0] * ext[1] * ext[2])
    return {"count": len(points), "volume": volume, "density": len(p) / volume if volume > 0 else 0,
            "normalized_points": p}


def check_volume_compatibility(volume1, volume2, threshold=0.3):
    diff = abs(volume1 - volume2) / max(volume1, volume2)
    return diff <= threshold, diff


def evaluate_clouds(X, Y, max_points=10000, volume_threshold=0.3):
    """evaluate_single_scene (reference :165-209) on two in-memory clouds."""
    a, b = analyze_point_cloud(X), analyze_point_cloud(Y)
    ok, vdiff = check_volume_compatibility(a["volume"], b["volume"], volume_threshold)
    if not ok:
        return None
    xn, yn = a["normalized_points"], b["normalized_points"]
    return {"mmd": compute_mmd_sampled(xn, yn, max_points), "cd": compute_chamfer_distance(xn, yn),
            "hd": compute_hausdorff_distance(xn, yn), "density_ratio": a["density"] / b["density"],
            "s3dis_points": len(X), "lidar_net_points": len(Y), "s3dis_density": a["density"],
            "lidar_net_density": b["density"], "s3dis_volume": a["volume"], "lidar_net_volume": b["volume"],
            "volume_diff": vdiff}

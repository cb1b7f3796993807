"""numpy restatement of the reference's engine post-processing.  TEST INFRASTRUCTURE ONLY.

Follows raycast_engine/raycast_engine_cpu.py line by line in meaning (not in text); the ray/triangle
cast itself (Open3D in the reference, :46-51) is delegated to the C oracle (oracle/lrc_oracle.c).
Pinned by tests/golden/engine_golden.npz, which was produced by running the reference's own
RaycastEngineCPU methods with the cast call substituted (tests/golden/make_engine_golden.py).
"""
import numpy as np

from .c_oracle import OracleMesh


def cast_dict(omesh: OracleMesh, rays, threads=1, brute=False):
    """The part of Open3D's cast_rays dict the path can observe: t_hit, primitive_ids, primitive_normals."""
    rays = rays.astype(np.float32)
    t, prim = omesh.brute(rays) if brute else omesh.cast(rays, threads=threads)
    return {"t_hit": t, "primitive_ids": prim, "primitive_normals": omesh.normals(prim)}


def rays_intersect_mesh(omesh: OracleMesh, rays, threads=1, brute=False, return_mask=False):
    """reference: RaycastEngineCPU.rays_intersect_mesh, raycast_engine_cpu.py:24-73."""
    if not isinstance(rays, np.ndarray):
        raise TypeError("rays must be a numpy array.")
    if rays.ndim != 2 or rays.shape[1] != 6:
        raise ValueError("rays must be a (N, 6) array.")
    rays = rays.astype(np.float32)                                    # :50
    depths = cast_dict(omesh, rays, threads, brute)["t_hit"]         # :51-53
    masks = depths != np.inf                                          # :54
    o, d = rays[:, :3], rays[:, 3:]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)                  # :57  float32
    finite = np.isfinite(depths)
    points = np.zeros_like(o)
    points[finite] = o[finite] + d[finite] * depths[finite, None]     # :60-62  mul then add, float32
    out = points[masks]                                               # :71  stable, ray order
    return (out, masks) if return_mask else out


def lidar_intersect_mesh(omesh: OracleMesh, lidar, threads=1, brute=False, return_index=False):
    """reference: RaycastEngineCPU.lidar_intersect_mesh, raycast_engine_cpu.py:75-111."""
    rays = lidar.get_rays()                                           # :91
    points, masks = rays_intersect_mesh(omesh, rays, threads, brute, return_mask=True)   # :92
    center = lidar.pose[:3, 3]                                        # :95  float64
    dist = np.linalg.norm(points - center, axis=1)                    # :96  float64
    near = dist < lidar.intrinsics.max_range                          # :97  strict
    points = points[near]
    if len(points) > 0:
        v = points - center
        v = v / np.linalg.norm(v, axis=1, keepdims=True)
        ang = np.degrees(np.arccos(np.abs(v[:, 2])))                  # :100-107
    else:
        ang = np.empty(0)                                             # :109
    if return_index:
        return points, ang, np.flatnonzero(masks)[near]               # surviving ray indices
    return points, ang

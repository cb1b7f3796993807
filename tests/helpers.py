"""Shared builders for the tests: sensors, poses, random rays, bit-pattern comparison."""
import dataclasses

import numpy as np


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64, 2: np.uint16}[a.dtype.itemsize])


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    if a.size == 0:
        return
    ne = bits(a) != bits(b)
    assert not ne.any(), f"{what}: {int(ne.sum())} of {ne.size} entries differ; first at {np.argwhere(ne)[0]}"


def pose(x, y, z, yaw=0.0):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    c, s = np.cos(yaw), np.sin(yaw)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def sensor_8x512():
    from lidar import Indoor8LineLidarIntrinsics
    return dataclasses.replace(Indoor8LineLidarIntrinsics.create_standard_8line(), horizontal_res=512)


def sensor_32x2048():
    from lidar import Indoor8LineLidarIntrinsics
    return dataclasses.replace(Indoor8LineLidarIntrinsics.create_dense_32line(), horizontal_res=2048)


def sensor_small(lines=4, width=96, max_range=20.0):
    from lidar import Indoor8LineLidarIntrinsics
    degs = list(np.linspace(25.0, -35.0, lines))
    return Indoor8LineLidarIntrinsics(vertical_res=lines, horizontal_res=width, max_range=max_range,
                                      vertical_degrees=degs)


def random_rays(n, lo, hi, seed=0, unit=True):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    if unit:
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)


def random_soup(n_tris, seed=0, extent=4.0, size=0.5):
    """Unstructured triangle soup (non-watertight, overlapping): stresses ties and traversal order."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, size=(n_tris, 1, 3))
    v = c + rng.normal(scale=size, size=(n_tris, 3, 3))
    return v.reshape(-1, 3), np.arange(3 * n_tris, dtype=np.int32).reshape(-1, 3)

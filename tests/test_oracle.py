"""CPU: the oracle against itself (brute force = the definition, vs its own BVH) and against
analytic known answers.  The Embree boundary itself is "parity unpinned" (oracle/lrc_oracle.c)."""
import numpy as np
import pytest

from helpers import assert_bit_equal, random_rays, random_soup
from oracle.c_oracle import OracleMesh


@pytest.mark.parametrize("seed,n_tris", [(0, 1), (1, 2), (2, 9), (3, 400), (4, 6000)])
def test_bvh_equals_brute_force_on_soups(seed, n_tris):
    v, f = random_soup(n_tris, seed)
    om = OracleMesh(v, f)
    rays = random_rays(1500, -5, 5, seed)
    tb, pb = om.brute(rays)
    t, p = om.cast(rays)
    assert_bit_equal(t, tb)
    assert_bit_equal(p, pb)
    t8, p8 = om.cast(np.concatenate([rays] * 3), threads=4)       # threaded path, same answers
    assert_bit_equal(t8[:1500], tb)
    assert_bit_equal(p8[3000:], pb)


def test_bvh_equals_brute_force_on_grid_room():
    from lidarcast import synth
    mesh = synth.make_room(size=(3, 2.5, 2), num_boxes=3, seed=11, cell=0.05)
    om = OracleMesh(mesh.vertices, mesh.triangles)
    rays = random_rays(1500, 0.2, 1.8, 5)
    tb, pb = om.brute(rays)
    t, p = om.cast(rays)
    assert_bit_equal(t, tb)
    assert_bit_equal(p, pb)
    assert np.isfinite(t).mean() > 0.99


def test_unit_cube_closed_form():
    from lidarcast import synth
    cube = synth.unit_cube()
    om = OracleMesh(cube.vertices, cube.triangles)
    rays = random_rays(4000, 0, 0, 1)
    t, prim = om.cast(rays)
    expect = 1.0 / np.abs(rays[:, 3:].astype(np.float64)).max(axis=1)
    assert np.isfinite(t).all() and np.abs(t - expect).max() < 1e-5
    n = om.normals(prim)
    k = np.abs(rays[:, 3:]).argmax(axis=1)                 # face axis
    assert np.allclose(np.abs(n[np.arange(len(n)), k]), 1.0)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0)


def test_contract_details():
    from lidarcast import synth
    q = synth.quad(z=2.0)
    om = OracleMesh(q.vertices, q.triangles)
    rays = np.array([[0.25, 0.25, 0, 0, 0, 1],       # hit at t = 2
                     [0.25, 0.25, 0, 0, 0, -1],      # behind
                     [5, 5, 0, 0, 0, 1],             # outside the quad
                     [0.25, 0.25, 0, 0, 0, 2],       # un-normalised direction: t is parametric
                     [0.25, 0.25, 2, 0, 0, 1],       # origin on the surface: tnear exclusive
                     [0.25, 0.25, 4, 0, 0, -1],      # back face: two-sided
                     [-2, 0, 2, 1, 0, 0],            # in-plane grazing ray: den = 0 -> miss
                     [0, 0, 0, 0, 0, 0]],            # zero direction -> miss
                    dtype=np.float32)
    t, prim = om.brute(rays)
    assert t[0] == 2.0 and t[3] == 1.0 and t[5] == 2.0
    assert np.isinf(t[[1, 2, 4, 6, 7]]).all() and (prim[[1, 2, 4, 6, 7]] == 0xFFFFFFFF).all()
    assert np.array_equal(om.normals(prim[:1]), [[0, 0, 1]])
    assert not om.normals(prim[1:2]).any()
    # diagonal shared edge: both triangles give the same t; the smaller row wins
    t, prim = om.brute(np.array([[0.5, 0.5, 0, 0, 0, 1]], dtype=np.float32))
    assert t[0] == 2.0 and prim[0] == 0
    # degenerate (zero-area) triangle never hits; empty mesh always misses
    om = OracleMesh(np.zeros((3, 3)), [[0, 1, 2]])
    assert np.isinf(om.brute(rays)[0]).all() and np.isinf(om.cast(rays)[0]).all()
    om = OracleMesh(np.zeros((0, 3)), np.zeros((0, 3)))
    assert np.isinf(om.cast(rays)[0]).all() and np.isinf(om.brute(rays)[0]).all()


def test_numpy_postprocessing_restatement():
    """o + (d/|d|)*t in float32, stable compaction, strict range filter, float64 incident angles."""
    from lidarcast import synth
    from lidar import create_lidar
    from oracle import np_oracle
    from helpers import pose, sensor_small
    mesh = synth.make_room(size=(3, 2.5, 2), num_boxes=2, seed=4, cell=0.1)
    om = OracleMesh(mesh.vertices, mesh.triangles)
    k = sensor_small(lines=5, width=64, max_range=1.4)
    lidar = create_lidar(k, pose(1.2, 1.1, 1.0, 0.3))
    rays = lidar.get_rays()
    pts, mask = np_oracle.rays_intersect_mesh(om, rays, return_mask=True)
    t, _ = om.cast(rays)
    assert pts.dtype == np.float32 and np.array_equal(mask, np.isfinite(t))
    d = rays[:, 3:] / np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    assert_bit_equal(pts, (rays[:, :3] + d * np.where(mask, t, 0)[:, None])[mask])
    p2, ang, idx = np_oracle.lidar_intersect_mesh(om, lidar, return_index=True)
    dist = np.linalg.norm(pts.astype(np.float64) - lidar.pose[:3, 3], axis=1)
    assert np.array_equal(idx, np.flatnonzero(mask)[dist < 1.4]) and 0 < len(idx) < mask.sum()
    assert ang.dtype == np.float64 and ang.min() >= 0 and ang.max() <= 90
    with pytest.raises(TypeError):
        np_oracle.rays_intersect_mesh(om, rays.tolist())
    with pytest.raises(ValueError):
        np_oracle.rays_intersect_mesh(om, rays[:, :5])

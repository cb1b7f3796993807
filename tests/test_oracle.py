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


def _mt_f64_numpy(v, f, rays):
    """Textbook two-sided Moeller-Trumbore over ALL triangles in float64 numpy: closest (t, row)."""
    v = v.astype(np.float32).astype(np.float64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    e1, e2 = b - a, c - a
    t_best = np.full(len(rays), np.inf)
    p_best = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    m_best = np.full(len(rays), -1.0)
    for i, r in enumerate(rays.astype(np.float64)):
        o, d = r[:3], r[3:]
        pv = np.cross(d, e2)
        det = (e1 * pv).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            s = o - a
            u = (s * pv).sum(1) * inv
            q = np.cross(s, e1)
            w = (q * d).sum(1) * inv
            t = (e2 * q).sum(1) * inv
        ok = (det != 0) & (u >= 0) & (w >= 0) & (u + w <= 1) & (t > 0) & np.isfinite(t)
        if ok.any():
            k = np.flatnonzero(ok)
            j = k[np.lexsort((k, t[k]))[0]]
            t_best[i], p_best[i] = t[j], j
            m_best[i] = min(u[j], w[j], 1 - u[j] - w[j])
    return t_best, p_best, m_best


def test_float64_witness_equals_numpy_brute_force():
    """orc_witness_f64 (double Moeller-Trumbore through the oracle's tree) == the same test over ALL triangles in
    numpy: the tree never hides a triangle from the witness."""
    from lidarcast import synth
    mesh = synth.make_room(size=(3, 2.5, 2), num_boxes=3, seed=11, cell=0.25)
    om = OracleMesh(mesh.vertices, mesh.triangles)
    rays = random_rays(600, 0.2, 1.8, 5)
    t64, p64, m64 = om.witness(rays, threads=2)
    tn, pn, mn = _mt_f64_numpy(mesh.vertices, mesh.triangles, rays)
    assert np.array_equal(np.isfinite(t64), np.isfinite(tn))
    h = np.isfinite(tn)
    assert np.abs(t64[h] - tn[h]).max() < 1e-12 and np.array_equal(p64[h], pn[h]) and np.abs(m64[h] - mn[h]).max() < 1e-9
    v, f = random_soup(300, 7)
    om = OracleMesh(v, f)
    rays = random_rays(400, -5, 5, 8)
    t64, p64, _ = om.witness(rays)
    tn, pn, _ = _mt_f64_numpy(v, f, rays)
    h = np.isfinite(tn)
    assert np.array_equal(np.isfinite(t64), h) and np.abs(t64[h] - tn[h]).max() < 1e-11 and np.array_equal(p64[h], pn[h])


def test_float32_definition_against_the_float64_witness():
    """The float32 hit definition (Embree's Moeller-Trumbore form + the box clause) against exact geometry on a
    tessellated room: same hit/miss decision, same triangle away from edges, |t32 - t64| <= 1e-5 m (the north-star
    tolerance on range), and the box clause never acts."""
    from lidarcast import synth
    from lidar import create_lidar
    from helpers import pose, sensor_small
    mesh = synth.make_room(size=(4, 3, 2.5), num_boxes=4, seed=5, cell=0.04)
    om = OracleMesh(mesh.vertices, mesh.triangles)
    k = sensor_small(lines=16, width=1024, max_range=30.0)
    rays = np.concatenate([create_lidar(k, pose(1.0 + 0.7 * i, 1.5, 1.1, 0.2 * i)).get_rays() for i in range(3)])
    t32, p32, rej = om.cast_diag(rays, threads=4)
    t64, p64, m64 = om.witness(rays, threads=4)
    assert int(rej.sum()) == 0                                   # the box clause never rejected a triangle
    h32, h64 = np.isfinite(t32), np.isfinite(t64)
    disagree = int((h32 != h64).sum())
    assert disagree <= 2, disagree                               # edge leaks in either direction: none expected
    both = h32 & h64
    assert np.abs(t32[both].astype(np.float64) - t64[both]).max() <= 1e-5
    other = both & (p32 != p64)
    assert (m64[other] < 1e-4).all()                             # a different triangle only on a shared edge
    assert other.sum() <= 10 and both.mean() > 0.99


def test_finite_ray_contract():
    from lidarcast import synth
    cube = synth.unit_cube()
    om = OracleMesh(cube.vertices, cube.triangles)
    rays = random_rays(64, 0, 0, 2)
    bad = rays.copy()
    bad[::4, 0] = np.nan
    bad[1::4, 4] = np.inf
    bad[2::4, 5] = -np.inf
    t, prim = om.cast(bad)
    tb, pb = om.brute(bad)
    sick = ~np.isfinite(bad).all(1)
    assert sick.sum() == 48 and np.isinf(t[sick]).all() and (prim[sick] == 0xFFFFFFFF).all()
    assert_bit_equal(t, tb)
    assert_bit_equal(prim, pb)
    assert np.isfinite(t[~sick]).all()
    assert np.isinf(om.witness(bad)[0][sick]).all()

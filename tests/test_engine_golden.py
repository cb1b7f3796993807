"""CPU: oracle/np_oracle.py (the numpy restatement of the reference's engine post-processing) against outputs of
the reference's own RaycastEngineCPU code captured by tests/golden/make_engine_golden.py (cast substituted).
-m gpu twin: the HIP engine against the same vectors (test_hip_engine_reproduces_reference_outputs)."""
import os

import numpy as np
import pytest

from conftest import REPO
from helpers import assert_bit_equal


@pytest.fixture(scope="module")
def eg():
    return np.load(os.path.join(REPO, "tests", "golden", "engine_golden.npz"))


class _Lidar:
    def __init__(self, eg, tag):
        from lidar import Indoor8LineLidarIntrinsics
        self.pose = eg[f"{tag}_pose"]
        self.intrinsics = Indoor8LineLidarIntrinsics(
            vertical_res=len(eg[f"{tag}_vertical_degrees"]), horizontal_res=int(eg[f"{tag}_width"]),
            max_range=float(eg[f"{tag}_max_range"]), vertical_degrees=list(eg[f"{tag}_vertical_degrees"]))
        self._rays = eg[f"{tag}_rays"]

    def get_rays(self):
        return self._rays


def _mesh(eg):
    from lidarcast.synth import TriangleMesh
    return TriangleMesh(eg["mesh_vertices"], eg["mesh_triangles"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_numpy_restatement_equals_reference_outputs(eg, tag):
    from lidar import IndoorLidar
    from oracle import np_oracle
    from oracle.c_oracle import OracleMesh
    om = OracleMesh(eg["mesh_vertices"], eg["mesh_triangles"])
    lid = _Lidar(eg, tag)
    # the build's own ray generator reproduces the rays the reference engine was fed
    assert_bit_equal(IndoorLidar(lid.intrinsics, lid.pose).get_rays(), lid.get_rays())
    t, _ = om.cast(lid.get_rays())
    assert_bit_equal(t, eg[f"{tag}_t_hit"])                       # same cast as at capture time
    assert_bit_equal(np_oracle.rays_intersect_mesh(om, lid.get_rays()), eg[f"{tag}_rays_intersect_points"])
    pts, ang = np_oracle.lidar_intersect_mesh(om, lid)
    assert_bit_equal(pts, eg[f"{tag}_lidar_points"])
    assert ang.dtype == np.float64 and ang.shape == eg[f"{tag}_lidar_angles"].shape
    assert_bit_equal(ang, eg[f"{tag}_lidar_angles"])
    if tag == "c":
        assert pts.shape == (0, 3) and ang.shape == (0,)


def test_float64_and_unnormalised_rays(eg):
    from oracle import np_oracle
    from oracle.c_oracle import OracleMesh
    om = OracleMesh(eg["mesh_vertices"], eg["mesh_triangles"])
    assert_bit_equal(np_oracle.rays_intersect_mesh(om, eg["d_rays64"]), eg["d_rays_intersect_points"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_hip_engine_reproduces_reference_outputs(eg, tag):
    from raycast_engine import RaycastEngineGPU
    eng = RaycastEngineGPU()
    mesh = _mesh(eg)
    lid = _Lidar(eg, tag)
    assert_bit_equal(eng.rays_intersect_mesh(rays=lid.get_rays(), mesh=mesh), eg[f"{tag}_rays_intersect_points"])
    pts, ang = eng.lidar_intersect_mesh(lid, mesh)
    assert_bit_equal(pts, eg[f"{tag}_lidar_points"])
    ref = eg[f"{tag}_lidar_angles"]
    assert ang.dtype == np.float64 and ang.shape == ref.shape
    if len(ref):
        assert np.abs(ang - ref).max() < 1e-9
    assert_bit_equal(eng.rays_intersect_mesh(rays=eg["d_rays64"], mesh=mesh), eg["d_rays_intersect_points"])
    with pytest.raises(TypeError):
        eng.rays_intersect_mesh(rays=eg["d_rays64"].tolist(), mesh=mesh)
    with pytest.raises(ValueError):
        eng.rays_intersect_mesh(rays=eg["d_rays64"][:, :5], mesh=mesh)

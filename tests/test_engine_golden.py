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


@pytest.mark.gpu
def test_metrics_reproduce_reference_values():
    """Row N3: Chamfer / Hausdorff / MMD on the GPU against values computed by the reference's own functions
    (tests/golden/make_metrics_golden.py), same seeded subsamples.  float32 distance matrices in the reference,
    float32 distances + float64 kernel sums here: tolerances 1e-6 relative (CD/HD), 1e-5 absolute (MMD)."""
    from lidarcast import metrics
    g = np.load(os.path.join(REPO, "tests", "golden", "metrics_golden.npz"))
    X, Y = g["X"], g["Y"]
    for tag, (a, b) in {"xy": (X, Y), "xx": (X, X), "small": (X[:2000], Y[:1500])}.items():
        np.random.seed(123)
        cd = metrics.compute_chamfer_distance(a, b)
        np.random.seed(124)
        hd = metrics.compute_hausdorff_distance(a, b)
        np.random.seed(125)
        mmd = metrics.compute_mmd_sampled(a, b, max_points=4000, gamma=1.0)
        assert abs(cd - g[f"{tag}_cd"]) <= 1e-6 * max(1.0, abs(g[f"{tag}_cd"])), (tag, cd, g[f"{tag}_cd"])
        assert abs(hd - g[f"{tag}_hd"]) <= 1e-6 * max(1.0, abs(g[f"{tag}_hd"])), (tag, hd, g[f"{tag}_hd"])
        assert abs(mmd - g[f"{tag}_mmd"]) <= 1e-5, (tag, mmd, g[f"{tag}_mmd"])
    s = metrics.analyze_point_cloud(X)
    assert abs(s["volume"] - g["X_volume"]) < 1e-4 * g["X_volume"] and abs(s["density"] - g["X_density"]) < 1e-4 * g["X_density"]
    ok, diff = metrics.check_volume_compatibility(s["volume"], metrics.analyze_point_cloud(Y)["volume"], 0.3)
    assert float(ok) == g["vol_compat"][0] and abs(diff - g["vol_compat"][1]) < 1e-6
    # brute-force numpy check of the two kernels on a small case
    a, b = X[:300], Y[:200]
    dm = np.linalg.norm(a[:, None] - b, axis=2)
    assert np.abs(metrics.min_distances(a, b) - dm.min(1)).max() < 1e-6
    d2 = np.maximum((a.astype(np.float64) ** 2).sum(1)[:, None] + (b.astype(np.float64) ** 2).sum(1)[None] -
                    2 * a.astype(np.float64) @ b.astype(np.float64).T, 0)
    assert abs(metrics.rbf_kernel_sum(a, b, 0.7) - np.exp(-0.7 * d2).sum()) < 1e-8 * d2.size
    with pytest.raises(ValueError):
        metrics.min_distances(a, np.zeros((0, 3), np.float32))


@pytest.mark.gpu
def test_evaluate_single_scene_module(tmp_path):
    """The reference's module name and call shape (evaluate_single_scene.py:15-23, :165-209): two PLY files in, the
    metric dictionary out; the same values as the in-memory functions; None for unreadable files or incompatible
    volumes."""
    import evaluate_single_scene as ess
    from containers import write_labeled_ply
    from lidarcast import metrics
    g = np.load(os.path.join(REPO, "tests", "golden", "metrics_golden.npz"))
    X, Y = g["X"][:3000].astype(np.float32), g["Y"][:2500].astype(np.float32)
    z8, z16 = np.zeros((len(X), 3), np.uint8), np.zeros(len(X), np.uint16)
    write_labeled_ply(tmp_path / "x.ply", X, z8, z16, z16)
    with open(tmp_path / "y.ply", "wb") as f:                       # Open3D-style cloud: double coordinates + colours
        rec = np.zeros(len(Y), dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("red", "u1"), ("green", "u1"),
                                      ("blue", "u1")])
        rec["x"], rec["y"], rec["z"] = Y[:, 0], Y[:, 1], Y[:, 2]
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\n"
                b"property double z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(Y))
        rec.tofile(f)
    px, py = ess.load_point_cloud(str(tmp_path / "x.ply")), ess.load_point_cloud(str(tmp_path / "y.ply"))
    assert px.dtype == np.float64 and np.array_equal(px, X.astype(np.float64)) and np.array_equal(py, Y.astype(np.float64))
    np.random.seed(5)
    got = ess.evaluate_single_scene(str(tmp_path / "x.ply"), str(tmp_path / "y.ply"), max_points=2000, volume_threshold=0.9)
    np.random.seed(5)
    want = metrics.evaluate_clouds(px, py, max_points=2000, volume_threshold=0.9)
    assert got == want and sorted(got) == sorted(
        ["mmd", "cd", "hd", "density_ratio", "s3dis_points", "lidar_net_points", "s3dis_density", "lidar_net_density",
         "s3dis_volume", "lidar_net_volume", "volume_diff"])
    assert got["s3dis_points"] == len(X) and got["lidar_net_points"] == len(Y)
    assert ess.evaluate_single_scene(str(tmp_path / "x.ply"), str(tmp_path / "nope.ply")) is None
    write_labeled_ply(tmp_path / "big.ply", X * 3.0, z8, z16, z16)
    assert ess.evaluate_single_scene(str(tmp_path / "x.ply"), str(tmp_path / "big.ply"), volume_threshold=0.3) is None

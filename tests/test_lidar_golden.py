"""CPU: the vectorised ray generators and parameter records against vectors captured from the
reference's own lidar/ package (tests/golden/make_golden.py)."""
import dataclasses
import hashlib
import json

import numpy as np
import pytest

from helpers import assert_bit_equal, sensor_32x2048, sensor_8x512
from lidar import (DualAxisLidar, DualAxisLidarIntrinsics, Indoor8LineLidarIntrinsics, IndoorLidar,
                   create_lidar)

POSES = ("identity", "translated", "yawed")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("pname", POSES)
def test_g1_multiline_rays(golden, pname):
    arrays, meta = golden
    m = arrays[f"pose_{pname}"]
    assert_bit_equal(IndoorLidar(sensor_8x512(), m).get_rays(), arrays[f"g1_8x512_{pname}"])
    r = IndoorLidar(sensor_32x2048(), m).get_rays()
    assert r.shape == (65536, 6) and r.dtype == np.float32
    assert_bit_equal(r[::97], arrays[f"g1_32x2048_{pname}_stride97"])
    assert sha(r) == meta[f"g1_32x2048_{pname}_sha256"]


@pytest.mark.parametrize("pname", POSES)
def test_g2_uniform_fov_branch(golden, pname):
    arrays, _ = golden
    k = dataclasses.replace(sensor_8x512(), vertical_degrees=None)
    assert_bit_equal(IndoorLidar(k, arrays[f"pose_{pname}"]).get_rays(), arrays[f"g2_uniform_8x512_{pname}"])


@pytest.mark.parametrize("seed,pname", [(0, "identity"), (1, "identity"), (12345, "identity"),
                                        (0, "yawed"), (7, "translated")])
def test_g3_dual_axis_seeded_stream(golden, seed, pname):
    arrays, meta = golden
    k = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
    np.random.seed(seed)
    r = DualAxisLidar(k, arrays[f"pose_{pname}"]).get_rays()
    tag = f"g3_seed{seed}_{pname}"
    assert list(r.shape) == meta[f"{tag}_shape"] and r.dtype == np.float32
    assert_bit_equal(r[:64], arrays[f"{tag}_head"])
    assert_bit_equal(r[-64:], arrays[f"{tag}_tail"])
    assert_bit_equal(r[::53], arrays[f"{tag}_stride53"])
    assert sha(r) == meta[f"{tag}_sha256"]
    assert float(np.random.random()) == meta[f"{tag}_next_uniform"]     # same number of draws consumed


def test_g3_stream_continues_across_poses(golden):
    arrays, meta = golden
    k = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
    np.random.seed(42)
    a = DualAxisLidar(k, arrays["pose_identity"]).get_rays()
    b = DualAxisLidar(k, arrays["pose_translated"]).get_rays()
    assert [list(a.shape), list(b.shape)] == meta["g3_seed42_two_poses_shapes"]
    assert [sha(a), sha(b)] == meta["g3_seed42_two_poses_sha256"]
    # an explicit RandomState gives the same stream without touching the global one
    rs = np.random.RandomState(42)
    a2 = DualAxisLidar(k, arrays["pose_identity"], rng=rs).get_rays()
    assert_bit_equal(a2, a)


def test_g4_factories_and_dispatch(golden):
    _, meta = golden
    for name, ref in meta["g4_factories"].items():
        if name in ("create_custom_lidar_args", "DualAxisLidarIntrinsics_default"):
            continue
        cls = DualAxisLidarIntrinsics if "dual" in name else Indoor8LineLidarIntrinsics
        obj = getattr(cls, name)()
        mine = json.loads(json.dumps(dataclasses.asdict(obj)))
        assert obj.get_total_points_per_scan() == ref["total_points_per_scan"]
        ref = {k: v for k, v in ref.items() if k not in ("total_points_per_scan", "range_limits")}
        assert mine == ref, name
    obj = Indoor8LineLidarIntrinsics.create_custom_lidar(num_beams=4, beam_angles=[10.0, 0.0, -10.0, -30.0],
                                                        horizontal_resolution=0.02, max_range=12.0)
    assert json.loads(json.dumps(dataclasses.asdict(obj))) == meta["g4_factories"]["create_custom_lidar_args"]
    assert json.loads(json.dumps(dataclasses.asdict(DualAxisLidarIntrinsics()))) == \
        meta["g4_factories"]["DualAxisLidarIntrinsics_default"]
    eye = np.eye(4)
    kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
    assert [type(create_lidar(sensor_8x512(), eye)).__name__, type(create_lidar(kd, eye)).__name__] == \
        meta["g4_create_lidar_types"]
    assert list(kd.get_range_limits()) == meta["g4_factories"]["create_blk2go_dual_axis"]["range_limits"]
    with pytest.raises(ValueError):
        create_lidar(object(), eye)
    with pytest.raises(AssertionError):
        IndoorLidar(sensor_8x512(), np.eye(3))


def test_sensor_direction_table_is_the_pose_free_part(golden):
    arrays, _ = golden
    k = sensor_8x512()
    dirs = IndoorLidar(k, np.eye(4)).sensor_directions()
    assert dirs.dtype == np.float64 and dirs.shape == (4096, 3)
    # identity / pure translation: world direction = float32(table) exactly (up to the sign of zero:
    # (-0)*1 + 0*0 is +0 in the reference's matrix product; the scan kernel evaluates the same sum)
    assert np.array_equal(dirs.astype(np.float32), arrays["g1_8x512_translated"][:, 3:])
    R = arrays["pose_translated"][:3, :3]
    world = ((dirs[:, 0:1] * R[:, 0] + dirs[:, 1:2] * R[:, 1]) + dirs[:, 2:3] * R[:, 2]).astype(np.float32)
    assert_bit_equal(world, arrays["g1_8x512_translated"][:, 3:])
    # rotated pose: the reference's np.dot is a BLAS product = fused chain fma(c,R2, fma(b,R1, a*R0)); the scan
    # kernel evaluates exactly that chain.  Emulated here with exact rational arithmetic on a sample.
    from fractions import Fraction
    R = arrays["pose_yawed"][:3, :3]
    ref = arrays["g1_8x512_yawed"][:, 3:]
    for i in range(0, len(dirs), 41):
        for j in range(3):
            inner = float(Fraction(dirs[i, 0]) * Fraction(R[j, 0]))
            inner = float(Fraction(dirs[i, 1]) * Fraction(R[j, 1]) + Fraction(inner))
            val = float(Fraction(dirs[i, 2]) * Fraction(R[j, 2]) + Fraction(inner))
            assert np.float32(val) == ref[i, j]
            assert val == float(np.dot(dirs[i:i + 1], R.T)[0, j])
    u = IndoorLidar(dataclasses.replace(k, vertical_degrees=None), np.eye(4)).sensor_directions()
    assert u.dtype == np.float64 and np.array_equal(u.astype(np.float32), arrays["g2_uniform_8x512_identity"][:, 3:])


def test_g5_waypoint_pose(golden):
    arrays, _ = golden
    from trajectory import Waypoint, poses_from_waypoints
    wps = [Waypoint(*w) for w in arrays["g5_waypoints"]]
    assert_bit_equal(poses_from_waypoints(wps), arrays["g5_pose_matrices"])


def test_g6_frame_container(golden):
    _, meta = golden
    from containers import S3DISSimFrame, ScanQuality
    q = ScanQuality(0.5, 3, 1.0, 0.1, 2.0, 3.0, 0.2)
    assert q.to_dict() == meta["g6_scan_quality_dict"]
    assert meta["g6_len_mismatch"] == "ValueError"
    with pytest.raises(ValueError):
        S3DISSimFrame(0, np.zeros((3, 3)), np.zeros(2), q)
    fr = S3DISSimFrame(5, np.arange(9, dtype=np.float32).reshape(3, 3), np.array([1.0, 2.0, 3.0]), q)
    ref = meta["g6_frame"]
    assert fr.get_num_points() == ref["num_points"] and fr.get_coverage_ratio() == ref["coverage"]
    assert fr.get_point_cloud_bounds() == ref["bounds"] and repr(fr) == ref["repr"]

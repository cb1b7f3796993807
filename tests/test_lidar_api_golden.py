"""Secondary lidar API (time-sampled dual-axis generators, noise helpers, generator entry points) against vectors
captured from the reference's own ``lidar`` package (tests/golden/make_lidar_api_golden.py).  Bit for bit, including
how much of the global numpy stream each call consumes."""
import dataclasses
import json
import os

import numpy as np
import pytest

from helpers import assert_bit_equal
from lidar import DualAxisLidar, DualAxisLidarIntrinsics, Indoor8LineLidarIntrinsics, IndoorLidar

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "lidar_api_golden.npz"))


@pytest.fixture(scope="module")
def META():
    with open(os.path.join(HERE, "golden", "lidar_api_golden.json")) as f:
        return json.load(f)


def small_sensor():
    return dataclasses.replace(DualAxisLidarIntrinsics.create_blk2go_dual_axis(), point_rate=5000)


def test_angles_at_time(G):
    kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
    np.random.seed(3)
    for t, line, phi, theta in G["angles_noisy"]:
        got = kd.calculate_angles_at_time(float(t), line_idx=int(line))
        assert_bit_equal(np.array(got, dtype=np.float64), np.array([phi, theta]))
    assert np.random.random() == G["angles_noisy_next_draw"][0]          # same number of draws consumed
    quiet = dataclasses.replace(kd, angle_noise_std=0.0)
    state = np.random.get_state()[1].copy()
    for t, line, phi, theta in G["angles_quiet"]:
        got = quiet.calculate_angles_at_time(float(t), line_idx=int(line))
        assert_bit_equal(np.array(got, dtype=np.float64), np.array([phi, theta]))
    assert np.array_equal(np.random.get_state()[1], state)               # noise off: the stream is untouched


def test_time_sequence(G, META):
    kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
    for name, k, fd in (("default", kd, None), ("small", small_sensor(), None), ("small_0p03", small_sensor(), 0.03)):
        ts = k.generate_time_sequence(fd)
        assert len(ts) == META[f"time_sequence_{name}_len"]
        assert_bit_equal(ts[:16], G[f"time_sequence_{name}_head"])
        assert_bit_equal(ts[-16:], G[f"time_sequence_{name}_tail"])


def test_time_sampled_generators(G, META):
    lidar = DualAxisLidar(intrinsics=small_sensor(), pose=G["pose_yawed"])
    np.random.seed(11)
    r = lidar.get_rays_frame()
    assert r.dtype == np.float32
    assert_bit_equal(r, G["rays_frame_small"])
    assert np.random.random() == G["rays_frame_small_next_draw"][0]
    np.random.seed(12)
    r, ts = lidar.get_spiral_scan_rays(num_points=257)
    assert_bit_equal(r, G["spiral_rays_257"])
    assert_bit_equal(ts, G["spiral_stamps_257"])
    np.random.seed(13)
    assert_bit_equal(lidar.get_rays_sequence(np.array([0.0, 0.001, 0.5, 2.25])), G["rays_sequence_custom"])
    np.random.seed(14)
    got = np.concatenate([lidar.get_rays_at_time(t) for t in (0.0, 0.0123, 0.77)])
    assert got.shape == (3, 6) and got.dtype == np.float32
    assert_bit_equal(got, G["rays_at_time"])
    np.random.seed(15)
    assert_bit_equal(lidar.add_noise_to_rays(G["noise_to_rays_in"]), G["noise_to_rays_out"])
    full = DualAxisLidar(intrinsics=DualAxisLidarIntrinsics.create_blk2go_dual_axis(), pose=G["pose_yawed"])
    assert full.get_total_rays() == META["get_total_rays"]


def test_custom_dual_axis_factory_fails_like_the_reference(META):
    assert META["create_custom_dual_axis"] == "TypeError"
    with pytest.raises(TypeError):
        DualAxisLidarIntrinsics.create_custom_dual_axis()


def test_add_noise(G):
    k8 = Indoor8LineLidarIntrinsics.create_standard_8line()
    ins = [G[f"add_noise_{t}"] for t in ("points", "ranges", "angles", "intens")]
    np.random.seed(21)
    for name, k in (("dropout", k8), ("nodrop", dataclasses.replace(k8, dropout_probability=0.0))):
        out = k.add_noise(*ins)
        for j, tag in enumerate(("points", "ranges", "angles", "intens")):
            assert_bit_equal(np.asarray(out[j]), G[f"add_noise_{name}_{tag}"])
    assert len(G["add_noise_dropout_points"]) < 200 == len(G["add_noise_nodrop_points"])


def test_generator_entry_points(G):
    o, d = IndoorLidar._gen_lidar_rays_with_vertical_degrees(pose=G["pose_yawed"], vertical_degrees=[12.5, 0.0, -7.25],
                                                              W=40)
    assert_bit_equal(o, G["gen_vdeg_o"])
    assert_bit_equal(d, G["gen_vdeg_d"])
    o, d = IndoorLidar._gen_lidar_rays(pose=G["pose_yawed"], fov_up=10.0, fov_down=25.0, H=5, W=33)
    assert_bit_equal(o, G["gen_uniform_o"])
    assert_bit_equal(d, G["gen_uniform_d"])

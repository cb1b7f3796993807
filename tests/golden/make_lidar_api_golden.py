#!/usr/bin/env python3
"""Golden vectors for the secondary lidar API (time-sampled dual-axis generators, noise helpers, the
multi-line generator entry points), captured from the REFERENCE's own ``lidar`` package (build container only).

    python tests/golden/make_lidar_api_golden.py      # writes tests/golden/lidar_api_golden.npz / .json

The fixture holds inputs and outputs only.
"""
import dataclasses
import json
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import lidar as _ref_lidar  # noqa: E402
from lidar import (DualAxisLidar, DualAxisLidarIntrinsics, Indoor8LineLidarIntrinsics, IndoorLidar)  # noqa: E402

assert os.path.realpath(_ref_lidar.__file__).startswith(REF), _ref_lidar.__file__


def pose(x, y, z, yaw):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    c, s = np.cos(yaw), np.sin(yaw)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def main():
    A, meta = {}, {"numpy": np.__version__}
    yawed = pose(1.25, -0.75, 1.0, 0.7)
    A["pose_yawed"] = yawed
    kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
    small = dataclasses.replace(kd, point_rate=5000)           # 500 samples per 0.1 s frame

    # calculate_angles_at_time: noisy (seeded) and noise-free, several lines
    np.random.seed(3)
    rows = []
    for t, line in ((0.0, 0), (0.0123, 0), (0.05, 7), (0.0999, 31), (0.31, 40), (1.7, 5)):
        phi, theta = kd.calculate_angles_at_time(t, line_idx=line)
        rows.append((t, line, phi, theta))
    A["angles_noisy"] = np.array(rows, dtype=np.float64)
    A["angles_noisy_next_draw"] = np.array([np.random.random()])
    quiet = dataclasses.replace(kd, angle_noise_std=0.0)
    A["angles_quiet"] = np.array([(t, line) + tuple(quiet.calculate_angles_at_time(t, line_idx=line))
                                  for t, line, _, _ in rows], dtype=np.float64)

    # generate_time_sequence
    for name, k, fd in (("default", kd, None), ("small", small, None), ("small_0p03", small, 0.03)):
        ts = k.generate_time_sequence(fd)
        A[f"time_sequence_{name}_head"] = ts[:16].copy()
        A[f"time_sequence_{name}_tail"] = ts[-16:].copy()
        meta[f"time_sequence_{name}_len"] = int(len(ts))

    # time-sampled generators, seeded global stream, yawed pose
    lidar = DualAxisLidar(intrinsics=small, pose=yawed)
    np.random.seed(11)
    A["rays_frame_small"] = lidar.get_rays_frame()
    A["rays_frame_small_next_draw"] = np.array([np.random.random()])
    np.random.seed(12)
    r, ts = lidar.get_spiral_scan_rays(num_points=257)
    A["spiral_rays_257"], A["spiral_stamps_257"] = r, ts
    np.random.seed(13)
    A["rays_sequence_custom"] = lidar.get_rays_sequence(np.array([0.0, 0.001, 0.5, 2.25]))
    np.random.seed(14)
    A["rays_at_time"] = np.concatenate([lidar.get_rays_at_time(t) for t in (0.0, 0.0123, 0.77)])
    np.random.seed(15)
    base = np.arange(600, dtype=np.float32).reshape(100, 6)
    A["noise_to_rays_in"] = base
    A["noise_to_rays_out"] = lidar.add_noise_to_rays(base)
    meta["get_total_rays"] = int(DualAxisLidar(intrinsics=kd, pose=yawed).get_total_rays())

    # create_custom_dual_axis: what does the reference do?
    try:
        DualAxisLidarIntrinsics.create_custom_dual_axis()
        meta["create_custom_dual_axis"] = "ok"
    except Exception as e:                                     # noqa: BLE001
        meta["create_custom_dual_axis"] = type(e).__name__

    # Indoor8LineLidarIntrinsics.add_noise, dropout on and off
    k8 = Indoor8LineLidarIntrinsics.create_standard_8line()
    rng = np.random.RandomState(5)
    pts = rng.uniform(-3, 3, size=(200, 3))
    rg, an, it = rng.uniform(0.2, 18, 200), rng.uniform(-0.3, 0.3, 200), rng.uniform(0, 1, 200)
    A["add_noise_points"], A["add_noise_ranges"], A["add_noise_angles"], A["add_noise_intens"] = pts, rg, an, it
    np.random.seed(21)
    for name, k in (("dropout", k8), ("nodrop", dataclasses.replace(k8, dropout_probability=0.0))):
        out = k.add_noise(pts, rg, an, it)
        for j, tag in enumerate(("points", "ranges", "angles", "intens")):
            A[f"add_noise_{name}_{tag}"] = np.asarray(out[j])

    # the multi-line generator entry points called directly
    o, d = IndoorLidar._gen_lidar_rays_with_vertical_degrees(pose=yawed, vertical_degrees=[12.5, 0.0, -7.25], W=40)
    A["gen_vdeg_o"], A["gen_vdeg_d"] = o, d
    o, d = IndoorLidar._gen_lidar_rays(pose=yawed, fov_up=10.0, fov_down=25.0, H=5, W=33)
    A["gen_uniform_o"], A["gen_uniform_d"] = o, d

    np.savez_compressed(os.path.join(OUT, "lidar_api_golden.npz"), **A)
    with open(os.path.join(OUT, "lidar_api_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(A), "arrays;", meta)


if __name__ == "__main__":
    main()

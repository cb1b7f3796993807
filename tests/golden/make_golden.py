#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE's own pure-numpy code (run in the build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz / *.json

Imports only the reference's ``lidar`` package, ``trajectory/trajectory_generator.py`` and
``containers/s3dis_sim_frame.py`` from /root/reference (read-only, never copied; the fixtures hold
inputs and outputs only).  The ray-triangle arithmetic itself (Open3D/Embree) is absent from the
reference tree and cannot be captured -- see oracle/lrc_oracle.c ("parity unpinned" at that boundary).
"""
import dataclasses
import hashlib
import importlib.util
import json
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
from lidar import (Indoor8LineLidarIntrinsics, DualAxisLidarIntrinsics, IndoorLidar,  # noqa: E402
                   DualAxisLidar, create_lidar)
import lidar as _ref_lidar  # noqa: E402

assert os.path.realpath(_ref_lidar.__file__).startswith(REF), _ref_lidar.__file__


def load_standalone(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pose(x, y, z, yaw):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    c, s = np.cos(yaw), np.sin(yaw)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


POSES = {"identity": pose(0, 0, 0, 0.0), "translated": pose(2.5, 2.0, 1.0, 0.0),
         "yawed": pose(1.25, -0.75, 1.0, 0.7)}


def main():
    meta = {"numpy": np.__version__}
    arrays = {}
    for k, m in POSES.items():
        arrays[f"pose_{k}"] = m

    # G1: multi-line sensor, vertical_degrees branch
    k8 = dataclasses.replace(Indoor8LineLidarIntrinsics.create_standard_8line(), horizontal_res=512)
    k32 = dataclasses.replace(Indoor8LineLidarIntrinsics.create_dense_32line(), horizontal_res=2048)
    for pname, m in POSES.items():
        r = IndoorLidar(intrinsics=k8, pose=m).get_rays()
        assert r.dtype == np.float32 and r.shape == (4096, 6)
        arrays[f"g1_8x512_{pname}"] = r
        r = IndoorLidar(intrinsics=k32, pose=m).get_rays()
        assert r.shape == (65536, 6)
        arrays[f"g1_32x2048_{pname}_stride97"] = r[::97].copy()
        meta[f"g1_32x2048_{pname}_sha256"] = sha(r)

    # G2: uniform-FOV branch
    ku = dataclasses.replace(k8, vertical_degrees=None)
    for pname, m in POSES.items():
        arrays[f"g2_uniform_8x512_{pname}"] = IndoorLidar(intrinsics=ku, pose=m).get_rays()

    # G3: dual-axis sensor with the global numpy stream seeded
    kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
    for seed, pname in ((0, "identity"), (1, "identity"), (12345, "identity"), (0, "yawed"),
                        (7, "translated")):
        np.random.seed(seed)
        r = DualAxisLidar(intrinsics=kd, pose=POSES[pname]).get_rays()
        tag = f"g3_seed{seed}_{pname}"
        meta[f"{tag}_shape"] = list(r.shape)
        meta[f"{tag}_sha256"] = sha(r)
        arrays[f"{tag}_head"] = r[:64].copy()
        arrays[f"{tag}_tail"] = r[-64:].copy()
        arrays[f"{tag}_stride53"] = r[::53].copy()
        # state of the global stream after the scan: the next draw pins the number of draws consumed
        meta[f"{tag}_next_uniform"] = float(np.random.random())
    # two consecutive poses on one stream (the simulator loop never reseeds)
    np.random.seed(42)
    a = DualAxisLidar(intrinsics=kd, pose=POSES["identity"]).get_rays()
    b = DualAxisLidar(intrinsics=kd, pose=POSES["translated"]).get_rays()
    meta["g3_seed42_two_poses_shapes"] = [list(a.shape), list(b.shape)]
    meta["g3_seed42_two_poses_sha256"] = [sha(a), sha(b)]

    # G4: factory parameter values
    fact = {}
    for name in ("create_standard_8line", "create_high_resolution_8line", "create_low_cost_8line",
                 "create_dense_32line", "create_leica_blk2go", "create_custom_lidar"):
        obj = getattr(Indoor8LineLidarIntrinsics, name)()
        fact[name] = dataclasses.asdict(obj)
        fact[name]["total_points_per_scan"] = obj.get_total_points_per_scan()
    obj = Indoor8LineLidarIntrinsics.create_custom_lidar(num_beams=4, beam_angles=[10.0, 0.0, -10.0, -30.0],
                                                        horizontal_resolution=0.02, max_range=12.0)
    fact["create_custom_lidar_args"] = dataclasses.asdict(obj)
    obj = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
    fact["create_blk2go_dual_axis"] = dataclasses.asdict(obj)
    fact["create_blk2go_dual_axis"]["total_points_per_scan"] = obj.get_total_points_per_scan()
    fact["create_blk2go_dual_axis"]["range_limits"] = list(obj.get_range_limits())
    fact["DualAxisLidarIntrinsics_default"] = dataclasses.asdict(DualAxisLidarIntrinsics())
    meta["g4_factories"] = fact
    meta["g4_create_lidar_types"] = [type(create_lidar(k8, POSES["identity"])).__name__,
                                     type(create_lidar(kd, POSES["identity"])).__name__]
    try:
        create_lidar(object(), POSES["identity"])
        meta["g4_create_lidar_bad"] = "no error"
    except ValueError as e:
        meta["g4_create_lidar_bad"] = "ValueError"

    # G5: Waypoint.to_pose_matrix
    tg = load_standalone("ref_trajectory_generator", "trajectory/trajectory_generator.py")
    wps = [(0.0, 0.0, 1.0, 0.0), (2.5, 2.0, 1.0, 0.7), (-1.0, 3.25, 0.5, -2.1), (4.0, 1.0, 1.0, np.pi)]
    arrays["g5_waypoints"] = np.array(wps)
    arrays["g5_pose_matrices"] = np.stack([tg.Waypoint(*w).to_pose_matrix() for w in wps])

    # G6: S3DISSimFrame / ScanQuality behaviour
    sf = load_standalone("ref_s3dis_sim_frame", "containers/s3dis_sim_frame.py")
    q = sf.ScanQuality(0.5, 3, 1.0, 0.1, 2.0, 3.0, 0.2)
    meta["g6_scan_quality_dict"] = q.to_dict()
    try:
        sf.S3DISSimFrame(0, np.zeros((3, 3)), np.zeros(2), q)
        meta["g6_len_mismatch"] = "no error"
    except ValueError:
        meta["g6_len_mismatch"] = "ValueError"
    fr = sf.S3DISSimFrame(5, np.arange(9, dtype=np.float32).reshape(3, 3), np.array([1.0, 2.0, 3.0]), q)
    meta["g6_frame"] = {"num_points": fr.get_num_points(), "coverage": fr.get_coverage_ratio(),
                        "bounds": fr.get_point_cloud_bounds(), "repr": repr(fr)}

    np.savez_compressed(os.path.join(OUT, "lidar_golden.npz"), **arrays)
    with open(os.path.join(OUT, "lidar_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(arrays), "arrays;", os.path.getsize(os.path.join(OUT, "lidar_golden.npz")), "bytes")


if __name__ == "__main__":
    main()

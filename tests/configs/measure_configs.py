#!/usr/bin/env python3
"""One-GPU measurements of the BASELINE.md configs other than the bench line (C1, C2, host-buffer C3, C4 slice).
Writes a JSON summary to stdout; DESIGN.md section 5 quotes it (profiles/r01_configs.json)."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
from lidar import DualAxisLidarIntrinsics, create_lidar  # noqa: E402
from lidarcast import synth  # noqa: E402
from raycast_engine import RaycastEngineGPU  # noqa: E402
from oracle import np_oracle  # noqa: E402
from oracle.c_oracle import OracleMesh  # noqa: E402

sys.path.insert(0, os.path.join(REPO, "tests"))
from helpers import pose, sensor_8x512  # noqa: E402


def med(f, n=7):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


out = {"host_threads": bench.host_threads()}
eng = RaycastEngineGPU()

# ---- C1 / C2: 8 x 512 single pose in synth_A1_office ----
a1 = synth.make_scene("synth_A1_office")
k8 = sensor_8x512()
lidar = create_lidar(k8, pose(4.0, 3.0, 1.0))
thr = bench.host_threads()


def c1_faithful():
    om = OracleMesh(a1.vertices, a1.triangles).build()
    np_oracle.lidar_intersect_mesh(om, lidar, threads=thr)
    om.free()


om1 = OracleMesh(a1.vertices, a1.triangles).build()
t = med(c1_faithful, 3)
out["C1_cpu_faithful_rays_per_s"] = 4096 / t
out["C1_cpu_faithful_ms"] = t * 1e3
t = med(lambda: np_oracle.lidar_intersect_mesh(om1, lidar, threads=thr))
out["C1_cpu_build_once_rays_per_s"] = 4096 / t
t0 = time.perf_counter()
scene = eng.scene_for(a1)
out["C2_scene_build_ms_total"] = (time.perf_counter() - t0) * 1e3
out["C2_scene_info"] = {k: v for k, v in scene.info.items() if not k.startswith("bounds")}
t = med(lambda: eng.lidar_intersect_mesh(lidar, a1), 15)
out["C2_hip_single_pose_ms_host_api"] = t * 1e3
out["C2_hip_rays_per_s_host_api"] = 4096 / t
pts, ang = eng.lidar_intersect_mesh(lidar, a1)
ref, _ = np_oracle.lidar_intersect_mesh(om1, lidar, threads=thr)
out["C2_bit_exact_points"] = bool(np.array_equal(pts.view(np.uint32), ref.view(np.uint32)))

# ---- C3 through the host-buffer API (PCIe + allocation inclusive) ----
a6 = synth.make_scene(bench.SCENE)
k32 = bench.c3_sensor()
poses = bench.c3_poses(0, 1)
t = med(lambda: eng.scan_poses(k32, poses, a6, want=("t", "prim", "normal3", "point3", "sem", "ins")), 5)
out["C3_host_api_rays_per_s"] = poses.shape[0] * 65536 / t
out["C3_host_api_ms"] = t * 1e3

# ---- C4 slice: BLK2GO dual axis, 8 poses on one GPU, host ray generation on the seeded stream ----
kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
np.random.seed(0)
p4 = [pose(1.0 + 0.02 * i, 3.0, 1.0) for i in range(8)]
t0 = time.perf_counter()
nrays = 0
for m in p4:
    lid = create_lidar(kd, m)
    rays = lid.get_rays()
    nrays += len(rays)
    eng.cast_rays(rays, a1, center=m[:3, 3], max_range=kd.max_range, want=("t", "point3", "sem", "ins"))
t = time.perf_counter() - t0
out["C4_slice_rays_per_s_1gpu_host_raygen"] = nrays / t
out["C4_slice_ms_per_pose"] = t / len(p4) * 1e3
print(json.dumps(out, indent=1))

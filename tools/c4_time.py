#!/usr/bin/env python3
"""tools/c4_time.py [poses] -- BASELINE config C4 (BLK2GO dual-axis sensor, np.random.seed(0), straight line through
synth_A1_office, 64 000 rays per pose before the 2 % dropout) through the plugin surface, three ways:
  host generator      RaycastEngineGPU.scan_lidars: rays from the vectorised host generator (bit-exact to the
                      reference's), one lrc_cast_segments launch, fixed-stride records back over PCIe   [round 1 path]
  host generator to frames   RaycastEngineGPU.scan_frames_lidars: the same rays, all of them at a fixed stride with the
                      dropout mask in page-locked memory, compaction in HBM (lrc_scan_rays_compact)   [the default now]
  device generator    RaycastEngineGPU.scan_frames_dual_axis: scan angles drawn on the host (the seeded stream), rays
                      formed in the kernel, compaction in HBM, kept rows into page-locked memory
  and the stages of the second: host RNG + angle formulae alone, the library call alone (angles already in page-locked
  memory), the device part alone (angles resident in HBM: lrc_scan_angles_dev + lrc_compact_dev, HIP events).
Prints one JSON object."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lidarcast  # noqa: E402
from lidar import DualAxisLidarIntrinsics, create_lidar  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidarcast._capi import LrcCompactIO  # noqa: E402
from raycast_engine import RaycastEngineGPU  # noqa: E402
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mesh = synth.make_scene("synth_A1_office")
eng = RaycastEngineGPU()
scene = eng.scene_for(mesh)
kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
poses = poses_from_waypoints(line_trajectory((1.0, 3.0, 1.0), (7.0, 3.0, 1.0), P))
N = 64000
out = {"poses": P, "rays_before_dropout": P * N}


def lidars():
    np.random.seed(0)
    return [create_lidar(kd, m) for m in poses]


for name, fn in (("host_generator", lambda: eng.scan_lidars(lidars(), mesh, want=("t", "point3", "sem", "ins"))),
                 ("host_generator_to_frames", lambda: eng.scan_frames_lidars(lidars(), mesh, want=("point3", "sem", "ins"))),
                 ("device_generator", lambda: eng.scan_frames_dual_axis(lidars(), mesh, want=("point3", "sem", "ins")))):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
        del r
    out[name] = {"seconds": min(ts[1:]), "rays_per_s": P * N / min(ts[1:])}

# stages of the device-generator path
t0 = time.perf_counter()
ls = lidars()
ang = eng.ctx.pinned.take(P * N * 16)[:P * N * 16].view(np.float64).reshape(P, N, 2)
keep = eng.ctx.pinned.take(P * N)[:P * N].reshape(P, N)
for i, l in enumerate(ls):
    phi, theta, k = l.scan_angles()
    ang[i, :, 0], ang[i, :, 1], keep[i] = phi, theta, k
out["host_angles_seconds"] = time.perf_counter() - t0
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    fr = scene.scan_angles_compact(poses, ang, keep, kd.max_range, want=("point3", "sem", "ins"))
    ts.append(time.perf_counter() - t0)
    total = fr["total"]
    del fr
out["library_call_seconds"] = min(ts[1:])
out["library_call_rays_per_s"] = P * N / min(ts[1:])
out["kept_rows"] = int(total)
dev = torch.device("cuda", 0)
d_poses = torch.from_numpy(poses.reshape(P, 16)).to(dev)
d_ang = torch.from_numpy(np.ascontiguousarray(ang)).to(dev)
d_keep = torch.from_numpy(np.ascontiguousarray(keep)).to(dev)
hits = lidarcast.DeviceHits(P * N, dev, want=("t", "prim", "point3", "sem", "ins", "tile_count"))
rows = torch.empty((P * N, 4), dtype=torch.float32, device=dev)
counts = torch.zeros(P, dtype=torch.int64, device=dev)
io = LrcCompactIO()
io.t, io.point3, io.sem, io.ins = (hits[a].data_ptr() for a in ("t", "point3", "sem", "ins"))
io.tile_count, io.counts, io.out_xyzl = hits["tile_count"].data_ptr(), counts.data_ptr(), rows.data_ptr()
st = torch.cuda.current_stream().cuda_stream
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
for a, b in ev:
    a.record()
    scene.scan_angles_dev(d_poses, d_ang.view(-1, 2), d_keep.view(-1), hits, kd.max_range, st)
    eng.ctx.compact_dev(P, N, io, st)
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev[2:])
out["device_resident_ms"] = ms[len(ms) // 2]
out["device_resident_rays_per_s"] = P * N / (ms[len(ms) // 2] * 1e-3)
assert int(counts.sum().item()) == total
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""tools/sorted_rays_bound.py -- what could re-packing rays by subtree bring, at best?

The two-phase idea (walk the shared head per wave, then re-pack the rays by the subtree they enter so that waves stay
coherent below the head) needs an extra pass over all rays.  Before building it, this measures its ceiling: the C3 rays
are cast as EXPLICIT rays (lrc_cast_dev, no in-kernel generation) in three orders --
  scanline   the order the sensor produces them (64 consecutive azimuths per wave): today's packing
  by-leaf    sorted by the BVH leaf slot of the triangle each ray finally hits, i.e. PERFECT knowledge of where every
             ray ends: no key a first phase could compute is better than this one
  shuffled   random order (what incoherent rays cost), for scale
and the kernel time of each order is printed (HIP events on the launch stream, median of 30)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lidarcast  # noqa: E402
from lidar import create_lidar  # noqa: E402
from lidarcast import synth  # noqa: E402

mesh = synth.make_scene(bench.SCENE)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
sensor = bench.c3_sensor()
poses = bench.c3_poses(0, 1)
rays = np.concatenate([create_lidar(sensor, m).get_rays() for m in poses])
n = len(rays)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
hits = lidarcast.DeviceHits(n, dev, want=("t", "prim", "normal3", "point3", "sem", "ins", "tile_count"))


def timed(r, label):
    d = torch.from_numpy(np.ascontiguousarray(r)).to(dev)
    for _ in range(3):
        scene.cast_dev(d, hits, center=poses[0][:3, 3], max_range=sensor.max_range, stream=st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record()
        scene.cast_dev(d, hits, center=poses[0][:3, 3], max_range=sensor.max_range, stream=st)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"{label:10s} {n} explicit rays: median {ms[15]:.4f} ms  min {ms[0]:.4f} ms  -> {n / ms[15] / 1e6:.2f} G rays/s")
    return ms[15]


t_scan = timed(rays, "scanline")
prim = hits["prim"].cpu().numpy().view(np.uint32)
_, slot_prim = scene.export_bvh()
slot_of = np.full(len(mesh.triangles) + 1, len(slot_prim), np.int64)
slot_of[slot_prim] = np.arange(len(slot_prim))
key = slot_of[np.minimum(prim, len(mesh.triangles))]           # misses sort to the end
order = np.argsort(key, kind="stable")
t_leaf = timed(rays[order], "by-leaf")
t_shuf = timed(rays[np.random.default_rng(0).permutation(n)], "shuffled")
print(f"ceiling of any re-packing by subtree: {t_scan / t_leaf:.2f}x on the trace alone, before the cost of the extra "
      f"pass (key + sort + gather of 24-byte rays + scatter of the 36-byte records, >= {n * (24 + 36 + 8) / 4e12 * 1e3:.3f} ms "
      f"at 4 TB/s)")

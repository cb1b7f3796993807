#!/usr/bin/env python3
"""tools/rebuild_time.py [W] -- the assembly kernels of the N-rank step ALONE on an idle GPU (no trace beside them): the rebuild
of W-1 remote slabs from triangle ids (lrc_cloud_from_prims_dev, own slab skipped), the rebuild from (t, label) pairs
(lrc_cloud_from_ranges_dev), the own-row scatter (lrc_compact_dev); microseconds per call and the algorithmic bytes moved
(4 or 8 B read + 16 B written per kept ray).  What is left of the N-rank step when these figures are subtracted is waiting."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidarcast._capi import LrcCompactIO  # noqa: E402
from lidarcast.distributed import PrimGather, RangeGather  # noqa: E402
from lidar import IndoorLidar  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(os.environ.get("REBUILD_STEPS", "30"))


class OneRank:
    @staticmethod
    def get_world_size(group=None):
        return 1


mesh = synth.make_scene(bench.SCENE)
if os.environ.get("REBUILD_SORTED") == "1":        # triangle rows in Morton order of their centroids: what slot-ordered ids would read
    cen = mesh.vertices[mesh.triangles].mean(axis=1)
    q = ((cen - cen.min(0)) / (cen.max(0) - cen.min(0) + 1e-9) * 1023).astype(np.uint64)

    def spread(v):
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")
    mesh = synth.TriangleMesh(vertices=mesh.vertices, triangles=np.ascontiguousarray(mesh.triangles[order]),
                              triangle_sem=mesh.triangle_sem[order], triangle_ins=mesh.triangle_ins[order])
    print("triangle rows sorted (Morton order of centroids)")
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
sensor = bench.c3_sensor()
poses = np.concatenate([bench.c3_poses(r, W) for r in range(W)])
P = len(poses) // W
dirs = IndoorLidar(sensor, np.eye(4)).sensor_directions()
N = len(dirs)
dev = torch.device("cuda", 0)
n = P * N
d_all = torch.from_numpy(poses.reshape(W * P, 16)).to(dev)
d_dirs = torch.from_numpy(dirs).to(dev)
st = torch.cuda.current_stream().cuda_stream
g = PrimGather(P, N, OneRank, dev, world=W)
rg = RangeGather(n, OneRank, dev, world=W)
tl = lidarcast.DeviceHits(0, dev, want=())
for v in range(W):
    tl.struct.prim = g.all_slabs[v * g.words:].data_ptr()
    tl.struct.tile_count = g.all_slabs[v * g.words + g.n:].data_ptr()
    tl.struct.t_label = rg.all_pairs[v * n:].data_ptr()
    scene.scan_poses_dev(d_all[v * P:(v + 1) * P], d_dirs, tl, sensor.max_range, st)
hits = lidarcast.DeviceHits(n, dev, want=("t", "point3", "sem", "ins", "tile_count"))
scene.scan_poses_dev(d_all[:P], d_dirs, hits, sensor.max_range, st)
cloud = torch.zeros((W * n, 4), dtype=torch.float32, device=dev)
counts = torch.zeros(W * P, dtype=torch.int64, device=dev)
own = LrcCompactIO()
own.t, own.point3, own.sem, own.ins = (hits[a].data_ptr() for a in ("t", "point3", "sem", "ins"))
io = LrcCompactIO()
io.t, io.point3, io.sem, io.ins = (hits[a].data_ptr() for a in ("t", "point3", "sem", "ins"))
io.tile_count, io.counts, io.out_xyzl = hits["tile_count"].data_ptr(), counts.data_ptr(), cloud.data_ptr()
torch.cuda.synchronize()


def timed(what, fn, rays, bytes_per_ray):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / K * 1e3
    print(f"W={W} {what:58s} {us:8.1f} us  {rays / us / 1e3:7.1f} G rays/s  {rays * bytes_per_ray / us / 1e6:6.2f} TB/s algorithmic", flush=True)


timed("rebuild from ids, all slabs", lambda: scene.cloud_from_prims_dev(
    d_all, d_dirs, g.all_prims, cloud, counts, g.all_tile_counts, poses_per_slab=P, slab_stride_bytes=g.stride_bytes, stream=st),
    W * n, 20)
if W > 1:
    timed("rebuild from ids, own slab scattered from the records", lambda: scene.cloud_from_prims_dev(
        d_all, d_dirs, g.all_prims, cloud, counts, g.all_tile_counts, poses_per_slab=P, slab_stride_bytes=g.stride_bytes, stream=st,
        own_slab=0, own_io=own), W * n, 20)
timed("rebuild from (t, label) pairs, all slabs", lambda: ctx.cloud_from_ranges_dev(d_all, d_dirs, rg.all_pairs, cloud, counts, st),
      W * n, 24)
timed("own rows: scan + scatter of one slab (lrc_compact_dev)", lambda: ctx.compact_dev(P, N, io, st), n, 36)
timed("memset of the cloud (16 B per ray written)", lambda: cloud.zero_(), W * n, 16)
src = torch.zeros((W * n, 4), dtype=torch.float32, device=dev)
timed("copy of the cloud (16 B read + 16 B written)", lambda: cloud.copy_(src), W * n, 32)

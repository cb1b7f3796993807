#!/usr/bin/env python3
"""tools/rebuild_time.py [W] -- stand-alone time of lrc_cloud_from_prims_dev over W ranks' slabs of the C3 scan
(no trace kernel next to it, no link): HIP events around 20 calls.  LRC_LIB selects a variant library."""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (puts the package on sys.path)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidar import IndoorLidar  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
mesh = synth.make_scene(bench.SCENE)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
sensor = bench.c3_sensor()
dirs = IndoorLidar(intrinsics=sensor, pose=np.eye(4)).sensor_directions()
poses = np.concatenate([bench.c3_poses(r, W) for r in range(W)])
P, N = bench.POSES_PER_GPU, len(dirs)
n = P * N
d_poses = torch.from_numpy(poses.reshape(-1, 16)).to(dev)
d_dirs = torch.from_numpy(dirs).to(dev)
words = n + n // 64
slabs = torch.full((W * words,), -1, dtype=torch.int32, device=dev)
hits = lidarcast.DeviceHits(n, dev, want=("t", "prim", "tile_count"))
st = torch.cuda.current_stream().cuda_stream
for v in range(W):
    hits.struct.prim = slabs[v * words:].data_ptr()
    hits.struct.tile_count = slabs[v * words + n:].data_ptr()
    scene.scan_poses_dev(d_poses[v * P:(v + 1) * P], d_dirs, hits, sensor.max_range, st)
pr = slabs.view(W, words)[:, :n].reshape(-1, 64)
print(f"lanes that start a run of equal triangle ids inside their tile: "
      f"{1.0 - float((pr[:, 1:] == pr[:, :-1]).float().mean()) * 63 / 64:.3f}", flush=True)
cloud = torch.empty((W * n, 4), dtype=torch.float32, device=dev)
counts = torch.zeros(W * P, dtype=torch.int64, device=dev)


def once(with_counts=True):
    scene.cloud_from_prims_dev(d_poses, d_dirs, slabs, cloud, counts, slabs[n:] if with_counts else None,
                               poses_per_slab=P, slab_stride_bytes=words * 4, stream=st)


for wc in (True, False):
    for _ in range(3):
        once(wc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        once(wc)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    k = int(counts.sum().item())
    print(f"W={W} with_counts={wc}: {ms*1e3:.1f} us per rebuild of {W*n} entries ({k} rows), "
          f"{(W*n*4 + k*16)/ms/1e6:.0f} GB/s (ids in + rows out)", flush=True)
k = int(counts.sum().item())
print("checksum", int(cloud[:k].view(torch.int32).to(torch.int64).sum().item()), k, flush=True)
# the 8-byte (t, label) payload for comparison: same scans, pairs written contiguously, rebuild without plane gathers
pairs = torch.empty((W * n, 2), dtype=torch.int32, device=dev)
hits2 = lidarcast.DeviceHits(n, dev, want=("t", "t_label"))
for v in range(W):
    hits2.struct.t_label = pairs[v * n:].data_ptr()
    scene.scan_poses_dev(d_poses[v * P:(v + 1) * P], d_dirs, hits2, sensor.max_range, st)
for _ in range(3):
    ctx.cloud_from_ranges_dev(d_poses, d_dirs, pairs, cloud, counts, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ctx.cloud_from_ranges_dev(d_poses, d_dirs, pairs, cloud, counts, st)
e1.record()
torch.cuda.synchronize()
k2 = int(counts.sum().item())
print(f"W={W} (t,label) pairs, counting pass included: {e0.elapsed_time(e1)/20*1e3:.1f} us per rebuild; checksum",
      int(cloud[:k2].view(torch.int32).to(torch.int64).sum().item()), k2, flush=True)
# reference point: plain fill of the same output
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    cloud.fill_(1.0)
e1.record()
torch.cuda.synchronize()
print(f"fill_ of the {cloud.numel()*4/1e6:.0f} MB cloud: {e0.elapsed_time(e1)/20*1e3:.1f} us")

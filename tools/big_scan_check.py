import sys, time; sys.path.insert(0, "."); import bench
import numpy as np, torch, lidarcast
from lidarcast import synth
from lidarcast._capi import LrcCompactIO
from lidar import IndoorLidar
from trajectory import line_trajectory, poses_from_waypoints
mesh = synth.make_scene(bench.SCENE); ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
k = bench.c3_sensor(); dirs = IndoorLidar(k, np.eye(4)).sensor_directions()
P = 1024; poses = poses_from_waypoints(line_trajectory((1.0,2.0,1.0),(4.0,2.0,1.0),P)); N = len(dirs); n = P*N
dev = torch.device("cuda",0)
hits = lidarcast.DeviceHits(n, dev, want=("t","prim","point3","sem","ins","tile_count"))
rows = torch.empty((n,4), dtype=torch.float32, device=dev); counts = torch.zeros(P, dtype=torch.int64, device=dev)
io = LrcCompactIO(); io.t, io.point3, io.sem, io.ins = (hits[a].data_ptr() for a in ("t","point3","sem","ins"))
io.tile_count, io.counts, io.out_xyzl = hits["tile_count"].data_ptr(), counts.data_ptr(), rows.data_ptr()
st = torch.cuda.current_stream().cuda_stream
dp, dd = torch.from_numpy(poses.reshape(P,16)).to(dev), torch.from_numpy(dirs).to(dev)
for _ in range(2):
    torch.cuda.synchronize(); t0=time.perf_counter()
    scene.scan_poses_dev(dp, dd, hits, k.max_range, st); ctx.compact_dev(P, N, io, st); torch.cuda.synchronize()
    dt=time.perf_counter()-t0
K = int(counts.sum().item())
t = hits["t"]
print(f"{n/1e6:.1f} M rays in {dt*1e3:.2f} ms = {n/dt/1e9:.2f} Grays/s; kept {K} ({K/n:.6f}); finite t {int(torch.isfinite(t).sum())}")
assert K == int(torch.isfinite(t).sum())
# spot-check last pose against the per-pose call
ref = scene.scan_poses(poses[-1:], dirs, k.max_range, want=("t","point3"))
assert np.array_equal(t[-N:].cpu().numpy().view(np.uint32), ref["t"].view(np.uint32))
keep = np.isfinite(ref["t"]); kk = int(keep.sum())
assert np.array_equal(rows[K-kk:K,:3].cpu().numpy().view(np.uint32), ref["point3"][keep].view(np.uint32))
print("ok")

#!/usr/bin/env python3
"""tools/wave_timeline.py [scene] [poses] -- when the waves of ONE trace launch start and end (C3 sensor).

Needs a measurement build of the library (the product kernel carries no measurement code):
    git apply tools/wave_clock.patch && tools/build_variant.sh waveclock && git apply -R tools/wave_clock.patch
    LRC_LIB=$PWD/build_variants/waveclock.so python3 tools/wave_timeline.py
In it every wave stamps s_memrealtime (100 MHz) at its first and last instruction and leaves both, with its XCD and its
hardware slot, in the intensity column.
Prints the distribution of wave lifetimes, the number of resident waves over the launch and what the ramp and the tail cost
against a launch that kept the steady-state rate from the first to the last nanosecond."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidar import Indoor8LineLidarIntrinsics, IndoorLidar  # noqa: E402
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402

scene_name = sys.argv[1] if len(sys.argv) > 1 else bench.SCENE
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lines, width = 32, 2048
mesh = synth.make_scene(scene_name)
Lx, Ly, _ = synth.scene_size(scene_name)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
k = Indoor8LineLidarIntrinsics(vertical_res=lines, horizontal_res=width, max_range=25.0,
                               vertical_degrees=list(np.linspace(15.0, -20.0, lines)))
dirs = IndoorLidar(k, np.eye(4)).sensor_directions()
poses = poses_from_waypoints(line_trajectory((1.0, Ly / 2, 1.0), (Lx - 1.0, Ly / 2, 1.0), P))
dev = torch.device("cuda", 0)
n = P * len(dirs)
hits = lidarcast.DeviceHits(n, dev, want=("t", "prim", "normal3", "point3", "sem", "ins", "tile_count", "intensity"))
d_poses, d_dirs = torch.from_numpy(poses.reshape(P, 16)).to(dev), torch.from_numpy(dirs).to(dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    scene.scan_poses_dev(d_poses, d_dirs, hits, k.max_range, st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); scene.scan_poses_dev(d_poses, d_dirs, hits, k.max_range, st); b.record()
torch.cuda.synchronize()
w = hits["intensity"].view(torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 64)
t0, t1, xcc = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64), w[:, 2] & 15
t1 = np.where(t1 < t0, t1 + (1 << 32), t1)
base = t0.min()
s, e = (t0 - base) * 0.01, (t1 - base) * 0.01          # microseconds
life = e - s
end = e.max()
print(f"{scene_name} 32x2048 x{P}: {len(s)} waves, HIP events {a.elapsed_time(b) * 1e3:.1f} us, first start -> last end {end:.1f} us")
q = np.percentile(life, [1, 10, 50, 90, 99, 99.9])
print("wave lifetime us: mean %.1f  p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f p99.9 %.1f max %.1f" % ((life.mean(),) + tuple(q) + (life.max(),)))
print(f"last wave starts at {s.max():.1f} us; waves still running after it: {(e > s.max()).sum()}; the 8 XCDs finish at "
      + " ".join(f"{e[xcc == x].max():.0f}" for x in range(8)) + " us; waves per XCD " + " ".join(str(int((xcc == x).sum())) for x in range(8)))
grid = np.linspace(0, end, 41)
res = [(int(((s <= t) & (e > t)).sum())) for t in grid]
print("resident waves at 40 instants: " + " ".join(str(r) for r in res))
done = np.sort(e)
for f in (0.5, 0.9, 0.99, 0.999, 1.0):
    print(f"  {f * 100:5.1f} % of the waves have ended by {done[min(len(done) - 1, int(f * len(done)) - 1)]:.1f} us")
# steady state: waves retired per microsecond between 25 % and 75 % of the launch
lo, hi = 0.25 * end, 0.75 * end
rate = ((e > lo) & (e <= hi)).sum() / (hi - lo)
print(f"steady-state retirement {rate:.1f} waves/us = {rate * 64 / 1e3:.2f} G rays/s; the whole launch at that rate would take "
      f"{len(s) / rate:.1f} us: ramp + tail cost {end - len(s) / rate:.1f} us")
late = np.argsort(e)[-8:]
print("the last 8 waves to end: " + "; ".join(f"tile {i} (pose {i // 1024}, line {(i % 1024) // 32}) {s[i]:.0f}->{e[i]:.0f}" for i in late))
# lifetime by start time: do late starters run faster?
for a_, b_ in ((0, 0.1), (0.1, 0.5), (0.5, 0.8), (0.8, 0.9), (0.9, 1.0)):
    m = (s >= a_ * s.max()) & (s <= b_ * s.max())
    print(f"  waves started in [{a_:.1f},{b_:.1f}] of the dispatch span: {int(m.sum())}, mean life {life[m].mean():.1f} us, max {life[m].max():.1f}")
# per scan line / per azimuth chunk: where do the long waves live?
tl = np.arange(len(s)) % 1024
ln, ch = tl // 32, tl % 32
print("per line: mean / p99 / max lifetime us")
for l in range(lines):
    m = ln == l
    print(f"  line {l:2d} ({k.vertical_degrees[l]:+6.1f} deg): {life[m].mean():5.1f} {np.percentile(life[m], 99):5.1f} {life[m].max():6.1f}")
print("per azimuth chunk of 64: mean / p99 / max lifetime us")
for c in range(32):
    m = ch == c
    print(f"  chunk {c:2d}: {life[m].mean():5.1f} {np.percentile(life[m], 99):5.1f} {life[m].max():6.1f}")
ps = np.arange(len(s)) // 1024
print("per pose: mean / max lifetime us: " + " ".join(f"{life[ps == q_].mean():.0f}/{life[ps == q_].max():.0f}" for q_ in range(P)))
# how predictable is a tile's lifetime from the same (line, chunk) of the previous pose?
L = life.reshape(P, 1024)
if P > 1:
    c_ = np.corrcoef(L[1:].ravel(), L[:-1].ravel())[0, 1]
    heavy = L > np.percentile(L, 99)
    print(f"correlation with the same tile of the previous pose: {c_:.3f}; of the 1 % longest waves, {100.0 * (heavy[1:] & heavy[:-1]).sum() / max(1, heavy[1:].sum()):.0f} % had a 1 %-longest predecessor")
# how long does a hardware wave slot stay empty between two waves?  (slot = XCD, shader engine/array, CU, SIMD, wave id)
hw = w[:, 3]
slot = (xcc.astype(np.int64) << 16) | (hw & 0x7FFF).astype(np.int64)
order = np.lexsort((s, slot))
so, ss, ee = slot[order], s[order], e[order]
same = so[1:] == so[:-1]
gap = (ss[1:] - ee[:-1])[same]
print(f"wave slots seen: {len(np.unique(slot))}; gap between a wave's last instruction and the next wave's first on the same slot: "
      f"mean {gap.mean():.2f} us, p10 {np.percentile(gap, 10):.2f} p50 {np.percentile(gap, 50):.2f} p90 {np.percentile(gap, 90):.2f} "
      f"(negative = slot ids reused before the stamp: {(gap < 0).sum()}); share of slot time empty between waves: "
      f"{gap.clip(min=0).sum() / (life.sum() + gap.clip(min=0).sum()):.3f}")

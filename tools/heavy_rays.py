#!/usr/bin/env python3
"""tools/heavy_rays.py [scene] -- which rays of the C3 scan are the expensive ones, and what do they look at?
Per-ray counters of the instrumented kernel (lrc_debug_scan_stats) for 8 poses of the trajectory, the scan's own hit
records beside them."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidar import IndoorLidar  # noqa: E402
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else bench.SCENE
mesh = synth.make_scene(name)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
sensor = bench.c3_sensor()
Lx, Ly, Lz = synth.scene_size(name)
allp = poses_from_waypoints(line_trajectory((1.0, Ly / 2, 1.0), (Lx - 1.0, Ly / 2, 1.0), 64))
sel = [5, 7, 20, 40, 58, 61, 62, 63]
poses = allp[sel]
dirs = IndoorLidar(intrinsics=sensor, pose=np.eye(4)).sensor_directions()
N = len(dirs)
st = scene.scan_stats(poses, dirs, sensor.max_range).reshape(len(sel), N, 5).astype(np.int64)
out = scene.scan_poses(poses, dirs, sensor.max_range, want=("t", "prim", "point3", "sem", "normal3"))
t = out["t"].reshape(len(sel), N)
sem = out["sem"].reshape(len(sel), N)
nrm = out["normal3"].reshape(len(sel), N, 3)
nodes, tris = st[..., 0], st[..., 1]
work = nodes + tris
print(f"{name}: room {Lx} x {Ly} x {Lz} m; per ray node steps mean {nodes.mean():.1f} max {nodes.max()}, triangle tests mean {tris.mean():.1f} max {tris.max()}")
flat = np.argsort(work.reshape(-1))[::-1][:40]
print("the 40 most expensive rays: pose line azimuth | node steps, triangle tests, wave-uniform steps, dead steps | t, label, |n.d| | direction")
for f in flat:
    p, i = divmod(int(f), N)
    line, az = divmod(i, 2048)
    d = dirs[i]
    c = abs(float(np.dot(nrm[p, i], d / np.linalg.norm(d)))) if np.isfinite(t[p, i]) else float("nan")
    print(f"  pose {sel[p]:2d} line {line:2d} az {az:4d} | {st[p, i, 0]:3d} {st[p, i, 1]:3d} {st[p, i, 2]:3d} {st[p, i, 3]:3d} | t {t[p, i]:7.3f} sem {sem[p, i]:2d} cos {c:5.3f} | {d[0]:+.3f} {d[1]:+.3f} {d[2]:+.3f}")
heavy = work >= np.percentile(work, 99.9)
print(f"top 0.1 % of rays (work >= {np.percentile(work, 99.9):.0f} steps+tests): misses {np.mean(~np.isfinite(t[heavy])):.2f} (all rays {np.mean(~np.isfinite(t)):.5f}); "
      f"mean t {np.nanmean(np.where(np.isfinite(t[heavy]), t[heavy], np.nan)):.2f} m (all {np.nanmean(np.where(np.isfinite(t), t, np.nan)):.2f}); "
      f"mean |cos incidence| {np.nanmean(np.abs(np.einsum('kj,kj->k', nrm[heavy], (dirs[np.nonzero(heavy)[1]] / np.linalg.norm(dirs[np.nonzero(heavy)[1]], axis=1, keepdims=True))))):.3f} "
      f"(all {np.nanmean(np.abs(np.einsum('pnj,nj->pn', nrm, dirs / np.linalg.norm(dirs, axis=1, keepdims=True)))):.3f}); labels of their hits {np.bincount(sem[heavy], minlength=14).tolist()} (all {np.round(np.bincount(sem.reshape(-1), minlength=14) / sem.size, 3).tolist()})")
# how does work grow with range and with grazing incidence?
fin = np.isfinite(t)
for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 6), (6, 9), (9, 25)):
    m = fin & (t >= lo) & (t < hi)
    if m.any():
        print(f"  t in [{lo},{hi}) m: {m.mean():6.3f} of rays, node steps {nodes[m].mean():5.1f}, triangle tests {tris[m].mean():5.1f}")
cosv = np.abs(np.einsum('pnj,nj->pn', nrm, dirs / np.linalg.norm(dirs, axis=1, keepdims=True)))
for lo, hi in ((0, 0.05), (0.05, 0.1), (0.1, 0.2), (0.2, 0.5), (0.5, 1.01)):
    m = fin & (cosv >= lo) & (cosv < hi)
    if m.any():
        print(f"  |cos incidence| in [{lo},{hi}): {m.mean():6.3f} of rays, node steps {nodes[m].mean():5.1f}, triangle tests {tris[m].mean():5.1f}")

# ---- the waves (64 consecutive azimuths) whose slowest lane is the most expensive, over the whole 64-pose scan ----
st_all = scene.scan_stats(allp, dirs, sensor.max_range).reshape(64, N, 5).astype(np.int64)
out_all = scene.scan_poses(allp, dirs, sensor.max_range, want=("t", "sem", "normal3"))
t_all, sem_all, nrm_all = out_all["t"].reshape(64, N), out_all["sem"].reshape(64, N), out_all["normal3"].reshape(64, N, 3)
w_all = (st_all[..., 0] + st_all[..., 1]).reshape(64, N // 64, 64)
wmax = w_all.max(2)
un = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
minabs = np.abs(un[:, :2]).min(1)            # how close to an axis direction in azimuth
order = np.argsort(wmax.reshape(-1))[::-1][:60]
print("the 60 waves with the most expensive lane: pose line chunk | its steps+tests (wave median) | that ray: az, min(|dx|,|dy|), |cos incidence|, label, t")
near_axis = 0
for f in order:
    p, wv = divmod(int(f), N // 64)
    lane = int(w_all[p, wv].argmax())
    i = wv * 64 + lane
    c = abs(float(np.dot(nrm_all[p, i], un[i])))
    near_axis += minabs[i] < 0.03
    print(f"  pose {p:2d} line {wv // 32:2d} chunk {wv % 32:2d} | {wmax[p, wv]:3d} ({int(np.median(w_all[p, wv])):2d}) | az {i % 2048:4d} min|d| {minabs[i]:.3f} cos {c:.3f} sem {sem_all[p, i]:2d} t {t_all[p, i]:.2f}")
print(f"of these 60, the expensive ray lies within 0.03 of an axis direction in azimuth: {near_axis}")
heavy_w = wmax >= np.percentile(wmax, 99.9)
hp, hw = np.nonzero(heavy_w)
lanes = w_all[hp, hw].argmax(1)
idx = hw * 64 + lanes
print(f"top 0.1 % of waves ({heavy_w.sum()}): slowest lane within 0.03 of an axis direction: {(minabs[idx] < 0.03).mean():.2f}, within 0.06: {(minabs[idx] < 0.06).mean():.2f}; "
      f"grazing hit (|cos| < 0.3): {np.mean([abs(float(np.dot(nrm_all[a, b], un[b]))) < 0.3 for a, b in zip(hp, idx)]):.2f}")
for eps in (0.01, 0.02, 0.03, 0.05):
    m = (minabs < eps).reshape(N // 64, 64).any(1)
    print(f"  waves that contain a ray within {eps} of an axis direction: {m.mean():.3f} of all waves; their mean max-lane work {wmax[:, m].mean():.1f} against {wmax[:, ~m].mean():.1f}; their share of the top 0.1 %: {heavy_w[:, m].sum() / heavy_w.sum():.2f}")

#!/usr/bin/env python3
"""tools/qnode_stress.py [scenes] [seed] -- quantised node images against the float32 nodes of the same tree over many random
scenes (the body of tests/test_configs_gpu.py::test_quantised_images_on_random_scenes, more of it); prints mismatches."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = lidarcast.Context(0)
used = bad = rays_total = hits_total = 0
for case in range(N):
    scale = 10.0 ** rng.uniform(-2.0, 2.5)
    T = int(rng.integers(1, 20000))
    centre = rng.uniform(-1.0, 1.0, 3) * scale * rng.choice([0.0, 0.5, 2.0])
    ext = np.array([1.0, rng.uniform(0.05, 1.0), rng.uniform(0.0 if case % 17 == 3 else 0.01, 1.0)])[rng.permutation(3)] * scale
    c = rng.uniform(-0.5, 0.5, (T, 1, 3)) * ext
    tri = (centre + c + rng.normal(scale=10.0 ** rng.uniform(-3, -1) * scale, size=(T, 3, 3)) * (ext > 0)).astype(np.float32)
    v, f = tri.reshape(-1, 3), np.arange(3 * T, dtype=np.int32).reshape(-1, 3)
    lo, hi = v.min(0).astype(np.float64), v.max(0).astype(np.float64)
    n = 64 * 800
    o = rng.uniform(lo - 0.3 * (hi - lo) - 1e-3 * scale, hi + 0.3 * (hi - lo) + 1e-3 * scale, (n, 3))
    onface = rng.random(n) < 0.2
    ax = rng.integers(0, 3, n)
    o[onface, ax[onface]] = np.where(rng.random(onface.sum()) < 0.5, lo[ax[onface]], hi[ax[onface]])
    pick = rng.integers(0, T, n)
    wgt = rng.dirichlet([1.0, 1.0, 1.0], n)                     # a point inside a random triangle: most rays hit something
    target = (tri[pick].astype(np.float64) * wgt[:, :, None]).sum(1)
    d = target - o
    kind = rng.integers(0, 8, n)
    d[kind == 0] = np.eye(3)[rng.integers(0, 3, (kind == 0).sum())] * rng.choice([-1.0, 1.0], ((kind == 0).sum(), 1))
    z = kind == 1
    d[z, rng.integers(0, 3, z.sum())] = rng.choice([0.0, -0.0], z.sum())
    t = kind == 2
    d[t, rng.integers(0, 3, t.sum())] = rng.choice([1e-38, -1e-38, 1e-42, -1e-30], t.sum())
    rays = np.concatenate([o, d], 1).astype(np.float32)
    os.environ["LRC_QNODES"] = "2"
    q = lidarcast.Scene(ctx, v, f)
    os.environ["LRC_QNODES"] = "0"
    w = lidarcast.Scene(ctx, v, f)
    used += q.info["quantised_nodes"]
    a = q.cast(rays, want=("t", "prim"))
    b = w.cast(rays, want=("t", "prim"))
    neq = int((a["t"].view(np.uint32) != b["t"].view(np.uint32)).sum() + (a["prim"] != b["prim"]).sum())
    rays_total += n
    hits_total += int(np.isfinite(a['t']).sum())
    if neq:
        bad += 1
        print(f"case {case}: {neq} differences (T={T}, scale={scale:.3g}, quantised={q.info['quantised_nodes']})")
    q.close(); w.close()
print(f"{N} scenes ({used} on the grid), {rays_total} rays, {hits_total} hits: {bad} scenes differ")

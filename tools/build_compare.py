#!/usr/bin/env python3
"""Device scene build against the host builder: every exported array byte for byte, and the build times.
Usage: build_compare.py [scene ...]   (names of lidarcast.synth.make_scene, default: a set of small and full-size meshes)"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401  (puts the package on sys.path)
import numpy as np  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402

ARRAYS = ("nodes", "tris", "slot_prim", "slot_label", "prim_plane", "nodes_q", "nodes_n")


def build(ctx, mesh, device, **env):
    old = {k: os.environ.get(k) for k in list(env) + ["LRC_DEVICE_BUILD"]}
    os.environ["LRC_DEVICE_BUILD"] = "1" if device else "0"
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        t0 = time.perf_counter()
        sc = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
        dt = (time.perf_counter() - t0) * 1e3
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return sc, dt


def compare(ctx, mesh, label, **env):
    h, th = build(ctx, mesh, False, **env)
    d, td = build(ctx, mesh, True, **env)
    d2, td2 = build(ctx, mesh, True, **env)          # second build: arena and pinned pages exist
    ih, idv = h.info, d.info
    ok = True
    for k in ("num_nodes", "num_leaves", "num_slots", "max_depth", "max_leaf_size", "bounds_lo", "bounds_hi",
              "quantised_nodes"):
        if ih[k] != idv[k]:
            print(f"  {label}: info.{k} differs: host {ih[k]} device {idv[k]}")
            ok = False
    assert idv["device_build"] == 1 and ih["device_build"] == 0
    for a in ARRAYS:
        x, y = h.export_array(a), d.export_array(a)
        if x.shape != y.shape or not np.array_equal(x, y):
            n = int((x[:min(len(x), len(y))] != y[:min(len(x), len(y))]).sum())
            first = int(np.flatnonzero(x[:min(len(x), len(y))] != y[:min(len(x), len(y))])[0]) if n else -1
            print(f"  {label}: array {a} differs ({len(x)} vs {len(y)} bytes, {n} bytes differ, first at {first})")
            ok = False
    print(f"{label}: T={ih['num_triangles']} nodes={ih['num_nodes']} depth={ih['max_depth']} "
          f"host {th:.1f} ms (build {ih['build_ms']:.1f}) | device first {td:.2f} ms, again {td2:.2f} ms "
          f"(upload {d2.info['upload_ms']:.2f} + build {d2.info['build_ms']:.2f}), infl {ih['leaf_inflation']:.4f}/"
          f"{idv['leaf_inflation']:.4f}  -> {'IDENTICAL' if ok else 'DIFFERENT'}")
    return ok


def main():
    ctx = lidarcast.Context(0)
    ok = True
    names = sys.argv[1:]
    if not names:
        rng = np.random.default_rng(3)
        small = synth.make_room(size=(3.0, 2.5, 2.0), num_boxes=3, seed=9, cell=0.05)
        ok &= compare(ctx, small, "small room")
        for ml in (1, 2, 3):
            ok &= compare(ctx, small, f"small room max_leaf={ml}", LRC_MAX_LEAF=ml)
        for bfs in (1, 2, 7, 100, 1000000):
            ok &= compare(ctx, small, f"small room bfs_nodes={bfs}", LRC_BFS_NODES=bfs)
        for sl in (0, 1, 5, -1):
            ok &= compare(ctx, small, f"small room depth_slack={sl}", LRC_DEPTH_SLACK=sl)
        ok &= compare(ctx, small, "small room median only", LRC_BUILD_MEDIAN_ONLY=1)

        class M:
            pass
        for nt in (5, 6, 9, 64, 65, 66, 200, 1024, 1025, 1030, 3000, 20000):
            m = M()
            c = rng.uniform(-3, 3, (nt, 1, 3))
            m.vertices = (c + rng.normal(scale=0.4, size=(nt, 3, 3))).reshape(-1, 3).astype(np.float32)
            m.triangles = np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3)
            m.triangle_sem = (np.arange(nt) % 13).astype(np.uint16)
            m.triangle_ins = None
            ok &= compare(ctx, m, f"soup {nt}")
            ok &= compare(ctx, m, f"soup {nt} median only", LRC_BUILD_MEDIAN_ONLY=1)
        # snapped coordinates: many equal centroids, zero extents, signed zeros
        m = M()
        nt = 5000
        v = np.round(rng.uniform(-2, 2, (nt, 3, 3)) * 2) / 2
        v[v == 0] = rng.choice([0.0, -0.0], size=int((v == 0).sum()))
        m.vertices = v.reshape(-1, 3).astype(np.float32)
        m.triangles = np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3)
        m.triangle_sem = None
        m.triangle_ins = None
        ok &= compare(ctx, m, "snapped soup")
        # all triangles identical: no SAH split anywhere
        m.vertices = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32), (3000, 1))
        m.triangles = np.arange(9000, dtype=np.uint32).reshape(-1, 3)
        ok &= compare(ctx, m, "3000 coincident triangles")
        names = ["synth_A6_office2", "synth_A1_office", "synth_rough_A6"]
    for name in names:
        mesh = synth.make_scene(name)
        ok &= compare(ctx, mesh, name)
        if name == "synth_A6_office2":
            ok &= compare(ctx, mesh, name + " median only", LRC_BUILD_MEDIAN_ONLY=1)
            ok &= compare(ctx, mesh, name + " depth_slack=0", LRC_DEPTH_SLACK=0)
    print("ALL IDENTICAL" if ok else "DIFFERENCES FOUND")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

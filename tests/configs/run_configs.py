#!/usr/bin/env python3
"""BASELINE.md configs C4 and C5 on the GPUs visible to this process group (1 GPU on the dev box).

C4: BLK2GO dual-axis sensor, np.random.seed(0) once, 256 poses on a straight line in synth_A1_office; the
    rays come from the host generator (seeded numpy stream, bit-identical to the reference), every rank casts
    its contiguous block of poses in one launch (lrc_cast_segments).
C5: the C3 sensor over synth_A1..A6, 64 poses each; aggregate rays/s + per-scene Chamfer distance between
    the HIP cloud and the CPU-oracle cloud of four poses per scene (definition of
    evaluate_single_scene.py:81-96 evaluated on the full clouds; 0.0 because the clouds are bit-identical).
Prints one JSON object (profiles/r01_c4_c5.json is a saved run).

Several GPUs: launch with ``python -m torch.distributed.run --nproc-per-node N tests/configs/run_configs.py``.  C4 then shards
the poses in contiguous blocks (every rank draws the whole seeded ray stream and keeps its block, so the rays do not
depend on N), casts its block, compacts it to 16-byte rows and joins ONE all-gather (lidarcast.distributed.CloudGather);
the SHA-256 of the assembled cloud is printed and is the same for every N.  C5 deals the scenes round-robin and
gathers the per-scene results.  LRC_DIST_BACKEND=gloo lets several ranks share one GPU for a rehearsal."""
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
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402


def chamfer(a, b):
    """mean(min_b |a-b|) + mean(min_a |b-a|), un-squared (evaluate_single_scene.py:81-96), on the FULL clouds:
    the reference draws two independent 5 000-point subsamples, which is non-zero even for identical clouds."""
    from scipy.spatial import cKDTree
    return float(cKDTree(b).query(a)[0].mean() + cKDTree(a).query(b)[0].mean())


def main():
    import hashlib
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("LRC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    out = {}
    eng = RaycastEngineGPU(device=local)
    quick = "--quick" in sys.argv

    # ---- C4 ----
    a1 = synth.make_scene("synth_A1_office")
    eng.scene_for(a1)
    kd = DualAxisLidarIntrinsics.create_blk2go_dual_axis()
    P = 32 if quick else 256
    poses = poses_from_waypoints(line_trajectory((1.0, 3.0, 1.0), (7.0, 3.0, 1.0), P))
    from lidarcast.distributed import CloudGather, shard_bounds
    b = shard_bounds(P, world)
    if world > 1:
        dist.barrier()
    np.random.seed(0)
    t0 = time.perf_counter()
    lidars = [create_lidar(kd, m) for m in poses]
    if world == 1:
        rec, off = eng.scan_lidars(lidars, a1, want=("t", "point3", "sem", "ins"))
    else:
        # one sequential stream for the whole trajectory: draw every pose's rays in order, keep this rank's block
        rays = [l.get_rays() for l in lidars]
        mine = list(range(b[rank], b[rank + 1]))

        class _Fixed:                 # a lidar whose rays were already drawn
            def __init__(self, l, r):
                self.pose, self.intrinsics, self._r = l.pose, l.intrinsics, r

            def get_rays(self):
                return self._r
        rec, off = eng.scan_lidars([_Fixed(lidars[i], rays[i]) for i in mine], a1, want=("t", "point3", "sem", "ins"))
    keep = np.isfinite(rec["t"])
    pts = rec["point3"][keep]
    lab = (rec["sem"][keep].astype(np.int32) | (rec["ins"][keep].astype(np.int32) << 16))
    per_pose = np.array([int(keep[off[i]:off[i + 1]].sum()) for i in range(len(off) - 1)], np.int64)
    if world > 1:
        max_local = int(max(b[1:] - b[:-1])) * kd.get_total_points_per_scan()
        g = CloudGather(max_local, int(max(b[1:] - b[:-1])), dist, dev)
        k = len(pts)
        g.slab[:k, :3] = torch.from_numpy(pts).to(dev)
        g.slab[:k, 3] = torch.from_numpy(lab).to(dev).view(torch.float32)
        g.counts[:len(per_pose)] = torch.from_numpy(per_pose).to(dev)
        g.gather()                                       # ONE all-gather for the whole scan
        P_all, L_all, C_all = g.assemble(torch.from_numpy(b[1:] - b[:-1]))
        pts, lab, per_pose = P_all.cpu().numpy(), L_all.cpu().numpy(), C_all.numpy()
        dist.barrier()
    t_all = time.perf_counter() - t0
    total_rays = sum(kd.get_total_points_per_scan() for _ in poses)
    out["C4"] = {"poses": P, "ranks": world, "rays_before_dropout": int(total_rays), "hits": int(len(pts)),
                 "seconds_total": t_all, "rays_per_s_incl_host_raygen": total_rays / t_all,
                 "cloud_sha256": hashlib.sha256(np.ascontiguousarray(pts).tobytes() + np.ascontiguousarray(lab).tobytes()
                                                + np.ascontiguousarray(per_pose).tobytes()).hexdigest(),
                 "note": "host ray generation (seeded numpy stream) + one lrc_cast_segments launch per rank through "
                         "the host-buffer API" + (" + one all-gather of the compacted rows" if world > 1 else "")
                         + "; the reference generator alone takes 1.9 s per pose"}
    # ray generation alone, to show where the time goes
    np.random.seed(0)
    t0 = time.perf_counter()
    for l in lidars[:16]:
        l.get_rays()
    out["C4"]["host_raygen_ms_per_pose"] = (time.perf_counter() - t0) / 16 * 1e3

    # ---- C5 ----
    from oracle import np_oracle
    from oracle.c_oracle import OracleMesh
    sensor = bench.c3_sensor()
    tot_rays, tot_t, scenes = 0, 0.0, {}
    for si, (name, spec) in enumerate(synth.SCENES.items()):
        if si % world != rank:
            continue                                   # scenes are dealt round-robin to the ranks
        mesh = synth.make_scene(name)
        Lx, Ly, _ = spec["size"]
        wps = line_trajectory((1.0, Ly / 2, 1.0), (Lx - 1.0, Ly / 2, 1.0), 8 if quick else 64)
        poses = poses_from_waypoints(wps)
        eng.scene_for(mesh)
        t0 = time.perf_counter()
        rec, n = eng.scan_poses(sensor, poses, mesh, want=("t", "point3"))
        dt = time.perf_counter() - t0
        cloud = rec["point3"][np.isfinite(rec["t"])]
        om = OracleMesh(mesh.vertices, mesh.triangles).build()
        sub = list(range(0, len(poses), max(1, len(poses) // 4)))[:4]
        ref = np.concatenate([np_oracle.lidar_intersect_mesh(om, create_lidar(sensor, poses[p]),
                                                             threads=bench.host_threads())[0] for p in sub])
        mine = np.concatenate([rec["point3"][p][np.isfinite(rec["t"][p])] for p in sub])
        scenes[name] = {"triangles": int(len(mesh.triangles)), "rays": int(poses.shape[0] * n),
                        "hits": int(len(cloud)), "host_api_seconds": dt,
                        "chamfer_vs_oracle": chamfer(mine, ref), "bit_identical_on_checked_poses":
                        bool(mine.shape == ref.shape and np.array_equal(mine.view(np.uint32), ref.view(np.uint32)))}
        tot_rays += poses.shape[0] * n
        tot_t += dt
        eng.clear_cache()
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (scenes, tot_rays, tot_t))
        scenes = {k: v for part in parts for k, v in part[0].items()}
        tot_rays = sum(part[1] for part in parts)
        tot_t = max(part[2] for part in parts)          # ranks work side by side: the slowest one sets the wall time
    out["C5"] = {"scenes": {k: scenes[k] for k in synth.SCENES if k in scenes}, "ranks": world,
                 "aggregate_rays_per_s_host_api": tot_rays / tot_t,
                 "note": f"{world} rank(s), host-buffer API (PCIe + allocation inclusive), scene build excluded"}
    if rank == 0:
        print(json.dumps(out, indent=1))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

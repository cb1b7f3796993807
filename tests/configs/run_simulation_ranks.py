#!/usr/bin/env python3
"""One rank of a torch.distributed job that runs S3DISSimulator.run_simulation (the rank-aware plugin surface).
usage (under ``python -m torch.distributed.run``): run_simulation_ranks.py <c4|multiline> <out.json>
Every rank hashes the scene it got back; rank 0 writes {"world", "frames", "sha256": [per rank], "quality"}.
LRC_DIST_BACKEND=gloo lets the ranks share one GPU (RCCL refuses two ranks on one device)."""
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402,F401  (puts the package on sys.path)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from lidarcast import synth  # noqa: E402
from s3dis_simulator import S3DISSimulator  # noqa: E402
from trajectory import line_trajectory  # noqa: E402


def main():
    what, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    backend = os.environ.get("LRC_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), rank=rank, world_size=world)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    cfg = {"raycast_engine": {"use_gpu": True}}
    if what == "c4":
        sim = S3DISSimulator(cfg, use_blk2go=True)
        sim.load_scene(synth.make_scene("synth_A1_office"), "synth_A1_office")
        wps = line_trajectory((1.0, 3.0, 1.0), (7.0, 3.0, 1.0), 256)
        np.random.seed(0)
    else:
        sim = S3DISSimulator(cfg, use_dense_lidar=True)
        sim.load_scene(synth.make_room(size=(5, 4, 3), num_boxes=6, seed=6, cell=0.04), "room")
        wps = line_trajectory((1.0, 2.0, 1.0), (4.0, 2.0, 1.0), 13, yaw=0.4)
    scene = sim.run_simulation(wps)          # picks up the process group: sharded scan, one all-gather
    counts = np.array([len(f.points) for f in scene.frames])
    h = hashlib.sha256()
    for a in (scene.combined_points(), *scene.combined_labels(), counts):
        h.update(np.ascontiguousarray(a).tobytes())
    parts = [None] * world
    dist.all_gather_object(parts, h.hexdigest())
    if rank == 0:
        q = [[f.scan_quality.num_points, float(f.scan_quality.range_mean)] for f in scene.frames][:3]
        with open(out, "w") as f:
            json.dump({"world": world, "frames": len(scene.frames), "sha256": parts, "quality": q}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A whole trajectory in one call: the reference's per-waypoint loop (s3dis_simulator.py:254-288) as
RaycastEngineGPU.scan_frames -- scan and compaction stay on the GPU, the kept rows of every pose arrive in page-locked
host memory, frames are views -- and as S3DISSimulator.run_simulation, whose ScanQuality statistics are computed on the
device with numpy's own summation order.

    python examples/scan_trajectory.py                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
           examples/scan_trajectory.py                       # waypoints sharded over 8 GPUs, same scene on every rank
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "indoor-point-cloud-datasets-controllable-generation-method-for-mobile-"
                                      "robots-3d-scene-perception_amd"))

import numpy as np  # noqa: E402
from lidarcast import synth  # noqa: E402
from s3dis_simulator import S3DISSimulator  # noqa: E402
from trajectory import line_trajectory, poses_from_waypoints  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:                                              # one process per GPU; "nccl" is RCCL on ROCm
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    backend = os.environ.get("LRC_DIST_BACKEND", "nccl")
    dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))

mesh = synth.make_room(size=(5.0, 4.0, 2.8), num_boxes=6, seed=2, cell=0.04)
sim = S3DISSimulator({"raycast_engine": {"use_gpu": True}}, use_dense_lidar=True)      # 32 lines x 4000 azimuths
sim.load_scene(mesh, "example_room")
waypoints = line_trajectory((1.0, 2.0, 1.0), (4.0, 2.0, 1.0), 16)

if world == 1:
    engine = sim.raycast_engine
    frames = engine.scan_frames(sim.lidar_config, poses_from_waypoints(waypoints), mesh,
                                want=("point3", "sem", "ins", "range_origin_stats"))
    per_pose = engine.split_frames(frames, "point3")
    print("scan_frames   :", len(per_pose), "frames,", frames["total"], "points; frame 3:", per_pose[3].shape, per_pose[3].dtype,
          "mean range %.4f m" % frames["range_origin_mean"][3])

t0 = time.perf_counter()
scene = sim.run_simulation(waypoints)                      # inside a distributed job: sharded scan, one all-gather
dt = time.perf_counter() - t0
q = scene.frames[3].scan_quality
print(f"run_simulation: {len(scene.frames)} frames, {scene.get_total_points()} points in {dt * 1e3:.1f} ms; "
      f"frame 3 coverage {q.coverage_ratio:.3f}, range {float(q.range_mean):.4f} +- {float(q.range_std):.4f} m")
assert q.range_mean == np.mean(np.linalg.norm(scene.frames[3].points, axis=1))       # numpy's bits
if world > 1:
    dist.barrier()
    dist.destroy_process_group()

"""Pose sharding across the GPUs of one node and assembly of the scene cloud.

The scan shards embarrassingly: waypoints are independent (reference loop, s3dis_simulator.py:254-288,
carries no state between iterations except the frame order).  Each rank scans a contiguous block of
poses against its own replica of the scene; one all-gather of the fixed-stride, locally compacted
cloud (RCCL over xGMI on GPUs, gloo in the CPU tests) assembles the scene point cloud in exactly the
order of ``np.vstack(frames)`` (containers/s3dis_sim_scene.py:326,362).
"""
import numpy as np


def shard_bounds(num_poses, world_size):
    """Contiguous pose blocks [b[r], b[r+1]); the first ``num_poses % world_size`` ranks get one more."""
    q, r = divmod(int(num_poses), int(world_size))
    sizes = [q + (1 if i < r else 0) for i in range(world_size)]
    b = np.zeros(world_size + 1, dtype=np.int64)
    b[1:] = np.cumsum(sizes)
    return b


def gather_cloud(local_points, local_labels, local_counts, max_local, dist, device=None):
    """All-gather a rank's compacted cloud.

    local_points (K_r,3) float32 / local_labels (K_r,) int32 (sem | ins<<16) / local_counts: per-pose
    hit counts of this rank (torch tensors).  Every rank contributes a fixed-size slab of ``max_local``
    rows (RCCL has no all-gather-v); the valid prefix lengths travel in a second, tiny all-gather.
    Returns (points (K,3), labels (K,), per_pose_counts (P,)) assembled in rank -> pose -> ray order.
    """
    import torch
    world = dist.get_world_size()
    dev = local_points.device if device is None else device
    k = int(local_points.shape[0])
    slab = torch.zeros((max_local, 4), dtype=torch.float32, device=dev)
    slab[:k, :3] = local_points
    slab[:k, 3] = local_labels.view(torch.float32) if local_labels.dtype == torch.int32 else local_labels
    out = torch.empty((world * max_local, 4), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, slab)
    npose = torch.tensor([local_counts.numel()], dtype=torch.int64, device=dev)
    nposes = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(nposes, npose)
    pmax = int(nposes.max().item())
    cpad = torch.zeros(pmax, dtype=torch.int64, device=dev)
    cpad[:local_counts.numel()] = local_counts.to(torch.int64)
    call = torch.empty(world * pmax, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(call, cpad)
    call = call.view(world, pmax).cpu()
    nposes = nposes.cpu()
    pts, labs, counts = [], [], []
    for r in range(world):
        c = call[r, :int(nposes[r])]
        kr = int(c.sum())
        seg = out[r * max_local:r * max_local + kr]
        pts.append(seg[:, :3])
        labs.append(seg[:, 3].contiguous().view(torch.int32))
        counts.append(c)
    return torch.cat(pts), torch.cat(labs), torch.cat(counts)

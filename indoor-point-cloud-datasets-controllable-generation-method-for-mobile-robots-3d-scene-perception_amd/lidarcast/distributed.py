"""Pose sharding across the GPUs of one node and assembly of the scene cloud.

The scan shards embarrassingly: waypoints are independent (reference loop, s3dis_simulator.py:254-288,
carries no state between iterations except the frame order).  Each rank scans a contiguous block of
poses against its own replica of the scene; one all-gather of the fixed-size slab of locally compacted
16-byte rows (x, y, z, label bits) -- RCCL over xGMI on GPUs, gloo in the CPU tests -- assembles the
scene point cloud in exactly the order of ``np.vstack(frames)`` (containers/s3dis_sim_scene.py:326,362).
"""
import numpy as np


def shard_bounds(num_poses, world_size):
    """Contiguous pose blocks [b[r], b[r+1]); the first ``num_poses % world_size`` ranks get one more."""
    q, r = divmod(int(num_poses), int(world_size))
    sizes = [q + (1 if i < r else 0) for i in range(world_size)]
    b = np.zeros(world_size + 1, dtype=np.int64)
    b[1:] = np.cumsum(sizes)
    return b


class CloudGather:
    """Reusable buffers for the per-scan all-gather (exactly one collective per scan).

    ``slab`` is this rank's (max_local + tail, 4) float32 send buffer.  Rows [0, max_local) receive the
    compacted cloud straight from the compaction kernel (lrc_compact_io.out_xyzl); rows past the rank's
    hit count are don't-care.  The tail rows hold the rank's per-pose hit counts as int64 (``counts`` is
    a view of them; lrc_compact_io.counts points there), so the prefix lengths travel inside the same
    collective.  Every rank contributes the same fixed size because RCCL has no all-gather-v.
    """

    def __init__(self, max_local, max_poses, dist, device):
        import torch
        self.dist, self.world = dist, dist.get_world_size()
        self.max_local, self.max_poses = int(max_local), int(max_poses)
        self.tail = (self.max_poses * 8 + 15) // 16
        self.rows = self.max_local + self.tail
        self.slab = torch.zeros((self.rows, 4), dtype=torch.float32, device=device)
        self.counts = self.slab[self.max_local:].view(-1).view(torch.int64)[:self.max_poses]
        self.all_rows = torch.empty((self.world * self.rows, 4), dtype=torch.float32, device=device)
        self.work = None

    def gather(self, async_op=False):
        """Enqueue the all-gather.  async_op=True returns at once; ``wait()`` before touching the buffers."""
        self.work = self.dist.all_gather_into_tensor(self.all_rows, self.slab, async_op=async_op)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    def assemble(self, poses_per_rank=None):
        """(points (K,3) f32, labels (K,) i32, per-pose counts) in rank -> pose -> ray order (host sync)."""
        import torch
        self.wait()
        per_rank = self.all_rows.view(self.world, self.rows, 4)
        call = per_rank[:, self.max_local:].reshape(self.world, -1).view(torch.int64)[:, :self.max_poses].cpu()
        pts, labs, counts = [], [], []
        for r in range(self.world):
            npose = self.max_poses if poses_per_rank is None else int(poses_per_rank[r])
            c = call[r, :npose]
            kr = int(c.sum())
            seg = per_rank[r, :kr]
            pts.append(seg[:, :3])
            labs.append(seg[:, 3].contiguous().view(torch.int32))
            counts.append(c)
        return torch.cat(pts), torch.cat(labs), torch.cat(counts)


class RangeGather:
    """The all-gather the pose-batched scan uses: 8-byte (t, label) pairs instead of 16-byte rows.

    A pose-batched scan is a pure function of (poses, direction table), which every rank holds, so a hit
    point can be rebuilt anywhere from its t (lrc_cloud_from_ranges_dev, bit-identical to the scan's own
    point).  ``slab`` (n, 2) int32 is the rank's send buffer -- the trace kernel writes its pairs straight
    into it (lrc_hits.t_label) -- and ``all_pairs`` (world*n, 2) receives every rank's pairs in rank order,
    i.e. in global pose order because ranks own contiguous pose blocks.  One collective per scan; halves the
    bytes on the xGMI links, which is what bounds the multi-GPU job."""

    def __init__(self, n_local, dist, device):
        import torch
        self.dist, self.world, self.n = dist, dist.get_world_size(), int(n_local)
        self.slab = torch.empty((self.n, 2), dtype=torch.int32, device=device)
        self.all_pairs = torch.empty((self.world * self.n, 2), dtype=torch.int32, device=device)
        self.work = None

    def gather(self, async_op=False):
        self.work = self.dist.all_gather_into_tensor(self.all_pairs, self.slab, async_op=async_op)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None


def gather_cloud(local_points, local_labels, local_counts, max_local, dist, device=None):
    """One-shot convenience form: all-gather a rank's compacted cloud given as separate tensors.
    Returns (points (K,3), labels (K,), per_pose_counts (P,)) assembled in rank -> pose -> ray order."""
    import torch
    dev = local_points.device if device is None else device
    world = dist.get_world_size()
    npose = torch.tensor([local_counts.numel()], dtype=torch.int64, device=dev)
    nposes = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(nposes, npose)
    g = CloudGather(max_local, int(nposes.max().item()), dist, dev)
    k = int(local_points.shape[0])
    g.slab[:k, :3] = local_points
    g.slab[:k, 3] = local_labels.view(torch.float32) if local_labels.dtype == torch.int32 else local_labels
    g.counts[:local_counts.numel()] = local_counts.to(torch.int64)
    g.gather()
    return g.assemble(nposes.cpu())

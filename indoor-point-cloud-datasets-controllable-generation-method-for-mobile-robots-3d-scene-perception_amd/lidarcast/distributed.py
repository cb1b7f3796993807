"""Pose sharding across the GPUs of one node and assembly of the scene cloud.

The scan shards embarrassingly: waypoints are independent (reference loop, s3dis_simulator.py:254-288,
carries no state between iterations except the frame order).  Each rank scans a contiguous block of
poses against its own replica of the scene; one all-gather per scan -- RCCL over xGMI on GPUs, gloo in the CPU
tests -- assembles the scene point cloud in exactly the order of ``np.vstack(frames)``
(containers/s3dis_sim_scene.py:326,362).  Three payloads, all fixed-stride (RCCL has no all-gather-v):
``PrimGather`` 4-byte triangle ids (the pose-batched scan; points rebuilt on the receiver), ``RangeGather`` 8-byte
(t, label) pairs (scans with range noise), ``CloudGather`` locally compacted 16-byte rows (host-generated rays).
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


class PrimGather:
    """The all-gather of the pose-batched scan at 4 bytes per ray: the hit triangle's row (lrc_hits.prim).

    A closest hit is a pure function of (pose, direction, triangle) and every rank holds the scene replica, the
    poses and the direction table, so the receiver recomputes t -- with the scan's own ray/triangle test, bit for
    bit -- and from it the point and the triangle's labels (lrc_cloud_from_prims_dev).  ``slab`` is the rank's send
    buffer: ``prim`` (poses_local * rays_per_pose) int32, written by the trace kernel itself (lrc_hits.prim points
    here), followed by ``tile_count`` (one int32 per 64 rays, lrc_hits.tile_count points here) so the receiver's
    rebuild needs no counting pass, padded to 16 bytes.  ``all_slabs`` receives every rank's slab in rank order =
    global pose order (ranks own contiguous pose blocks of ``poses_local`` poses; a rank with fewer poses pads with
    -1 ids).  One collective per scan, half the bytes of RangeGather on the xGMI links."""

    def __init__(self, poses_local, rays_per_pose, dist, device, world=None):
        import torch
        self.dist = dist
        self.world = dist.get_world_size() if world is None else int(world)
        self.poses_local, self.rays_per_pose = int(poses_local), int(rays_per_pose)
        self.n = self.poses_local * self.rays_per_pose
        self.fused_counts = self.rays_per_pose % 64 == 0
        self.ntiles = self.poses_local * (self.rays_per_pose // 64) if self.fused_counts else 0
        self.words = (self.n + self.ntiles + 3) // 4 * 4
        self.stride_bytes = self.words * 4
        self.slab = torch.full((self.words,), -1, dtype=torch.int32, device=device)
        self.prim = self.slab[:self.n]
        self.tile_count = self.slab[self.n:self.n + self.ntiles] if self.fused_counts else None
        if self.fused_counts:
            self.tile_count.zero_()
        self.all_slabs = torch.empty((self.world * self.words,), dtype=torch.int32, device=device)
        # ``world`` may exceed the process group (single-GPU diagnostics): the collective fills the leading slabs
        self.recv = self.all_slabs[:dist.get_world_size() * self.words]
        self.work = None

    @property
    def all_prims(self):
        """Pointer view for lrc_cloud_from_prims_dev: entry 0 of slab 0."""
        return self.all_slabs

    @property
    def all_tile_counts(self):
        return self.all_slabs[self.n:] if self.fused_counts else None

    def gather(self, async_op=False):
        self.work = self.dist.all_gather_into_tensor(self.recv, self.slab, async_op=async_op)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    def per_rank(self):
        """Host view for tests: list of (prim (poses_local, rays_per_pose), tile_count or None) per rank."""
        self.wait()
        a = self.all_slabs.view(self.world, self.words).cpu().numpy()
        return [(a[r, :self.n].reshape(self.poses_local, self.rays_per_pose),
                 a[r, self.n:self.n + self.ntiles] if self.fused_counts else None) for r in range(self.world)]


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

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

    def __init__(self, max_local, max_poses, dist, device, group=None, world=None):
        import torch
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.max_local, self.max_poses = int(max_local), int(max_poses)
        self.tail = (self.max_poses * 8 + 15) // 16
        self.rows = self.max_local + self.tail
        self.slab = torch.zeros((self.rows, 4), dtype=torch.float32, device=device)
        self.counts = self.slab[self.max_local:].view(-1).view(torch.int64)[:self.max_poses]
        self.all_rows = torch.empty((self.world * self.rows, 4), dtype=torch.float32, device=device)
        # ``world`` may exceed the process group (single-GPU diagnostics): the collective fills the leading slabs
        self.recv = self.all_rows[:dist.get_world_size(group) * self.rows]
        self.work = None

    def gather(self, async_op=False):
        """Enqueue the all-gather.  async_op=True returns at once; ``wait()`` before touching the buffers."""
        self.work = self.dist.all_gather_into_tensor(self.recv, self.slab, group=self.group, async_op=async_op)

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

    def __init__(self, n_local, dist, device, world=None):
        import torch
        self.dist, self.n = dist, int(n_local)
        self.world = dist.get_world_size() if world is None else int(world)
        self.slab = torch.empty((self.n, 2), dtype=torch.int32, device=device)
        self.all_pairs = torch.empty((self.world * self.n, 2), dtype=torch.int32, device=device)
        # ``world`` may exceed the process group (single-GPU diagnostics): the collective fills the leading slabs
        self.recv = self.all_pairs[:dist.get_world_size() * self.n]
        self.work = None

    def gather(self, async_op=False):
        self.work = self.dist.all_gather_into_tensor(self.recv, self.slab, async_op=async_op)

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

    def __init__(self, poses_local, rays_per_pose, dist, device, world=None, group=None):
        import torch
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if world is None else int(world)
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
        self.recv = self.all_slabs[:dist.get_world_size(group) * self.words]
        self.work = None

    @property
    def all_prims(self):
        """Pointer view for lrc_cloud_from_prims_dev: entry 0 of slab 0."""
        return self.all_slabs

    @property
    def all_tile_counts(self):
        return self.all_slabs[self.n:] if self.fused_counts else None

    def gather(self, async_op=False):
        self.work = self.dist.all_gather_into_tensor(self.recv, self.slab, group=self.group, async_op=async_op)

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


# ---- the plugin surface on N ranks -------------------------------------------------------------------------------
def active_group(process_group=None):
    """(dist module, group) when this process is a rank of an initialised torch.distributed job with more than one
    rank (or ``process_group`` is given), else (None, None).  torch is not imported for single-process callers."""
    import sys
    if process_group is None and "torch" not in sys.modules:
        return None, None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None, None
    if dist.get_world_size(process_group) <= 1:
        return None, None
    return dist, process_group


def scan_frames_sharded(engine, intrinsics, poses, mesh, dist, group=None):
    """``RaycastEngineGPU.scan_frames`` for a trajectory sharded over the ranks of a process group: contiguous pose
    blocks (shard_bounds), every rank scans its block against its scene replica, ONE all-gather of the hit triangle
    ids (PrimGather, 4 B per ray) and every rank rebuilds the whole scene cloud, so every rank returns the frames of
    ALL poses -- the same bytes as a single-process scan, whatever the number of ranks
    (reference loop: s3dis_simulator.py:254-288; assembly order: containers/s3dis_sim_scene.py:326).

    The engine supplies three device steps (RaycastEngineHIP implements them with the HIP kernels; the CPU tests
    plug in a stand-in so that sharding, gather and ordering run under gloo without a GPU):
      engine.prim_gather(poses_local, rays_per_pose, dist, group) -> PrimGather
      engine.scan_block_into(gather, intrinsics, block_poses, mesh)       trace -> ids + keep counts in the send slab
      engine.cloud_from_gather(gather, intrinsics, padded_poses, mesh)    -> (rows (K,4) f32 numpy, counts numpy
                                                                              [, {"range_origin_mean", "range_origin_std"}
                                                                               per-pose statistics computed on the device])
    """
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4, 4)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    P = len(poses)
    b = shard_bounds(P, world)
    per = int(max(b[1:] - b[:-1])) if P else 0
    if per == 0:
        return {"point3": np.zeros((0, 3), np.float32), "sem": np.zeros(0, np.uint16), "ins": np.zeros(0, np.uint16),
                "counts": np.zeros(0, np.int64), "total": 0}
    n_rays = len(engine._direction_table(intrinsics))
    g = engine.prim_gather(per, n_rays, dist, group)
    engine.scan_block_into(g, intrinsics, poses[b[rank]:b[rank + 1]], mesh)
    g.gather()                                   # the ONE collective of the scan
    padded = np.tile(np.eye(4), (world * per, 1, 1))
    for r in range(world):
        padded[r * per:r * per + (b[r + 1] - b[r])] = poses[b[r]:b[r + 1]]
    res = engine.cloud_from_gather(g, intrinsics, padded, mesh)
    rows, counts = res[0], res[1]
    stats = res[2] if len(res) > 2 else {}
    real = np.concatenate([np.arange(r * per, r * per + (b[r + 1] - b[r])) for r in range(world)]).astype(np.int64)
    counts = np.asarray(counts, dtype=np.int64)
    assert int(counts.sum()) == int(counts[real].sum()), "a padded pose produced returns"
    lab = np.ascontiguousarray(rows[:, 3]).view(np.uint32)
    out = {"point3": np.ascontiguousarray(rows[:, :3]), "sem": (lab & 0xFFFF).astype(np.uint16),
           "ins": (lab >> 16).astype(np.uint16), "counts": counts[real], "total": int(len(rows))}
    for k, v in stats.items():
        out[k] = np.asarray(v)[real]
    return out


def scan_lidars_sharded(engine, lidars, mesh, dist, group=None, device=None):
    """Host-generated rays (dual-axis sensor) on N ranks.  Every rank draws the rays of EVERY pose, in pose order, from
    the one numpy stream -- so the rays do not depend on N and the stream ends where a single process would leave it
    -- casts its contiguous block of poses in one launch, compacts it, and joins ONE all-gather of the 16-byte rows
    (CloudGather).  Returns the frames of all poses (same dict as scan_frames_sharded, plus incident_deg=None)."""
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    P = len(lidars)
    b = shard_bounds(P, world)
    rays = [l.get_rays() for l in lidars]              # the whole trajectory's stream, on every rank

    class _Drawn:                                       # a sensor whose rays were already drawn
        def __init__(self, l, r):
            self.pose, self.intrinsics, self._r = l.pose, l.intrinsics, r

        def get_rays(self):
            return self._r
    mine = [_Drawn(lidars[i], rays[i]) for i in range(b[rank], b[rank + 1])]
    per = int(max(b[1:] - b[:-1]))
    max_local = per * max(len(r) for r in rays) if rays else 0
    if mine:
        rec, off = engine.scan_lidars(mine, mesh, want=("t", "point3", "sem", "ins"))
        keep = np.isfinite(rec["t"])
        pts = rec["point3"][keep]
        lab = rec["sem"][keep].astype(np.int32) | (rec["ins"][keep].astype(np.int32) << 16)
        cnt = np.array([int(keep[off[i]:off[i + 1]].sum()) for i in range(len(off) - 1)], np.int64)
    else:
        pts, lab, cnt = np.zeros((0, 3), np.float32), np.zeros(0, np.int32), np.zeros(0, np.int64)
    dev = device if device is not None else engine.torch_device()
    g = CloudGather(max(max_local, 1), max(per, 1), dist, dev, group=group)
    k = len(pts)
    if k:
        g.slab[:k, :3] = torch.from_numpy(pts).to(dev)
        g.slab[:k, 3] = torch.from_numpy(lab).to(dev).view(torch.float32)
    g.counts.zero_()
    if len(cnt):
        g.counts[:len(cnt)] = torch.from_numpy(cnt).to(dev)
    g.gather()                                          # the ONE collective of the scan
    P_all, L_all, C_all = g.assemble(torch.from_numpy(b[1:] - b[:-1]))
    lab_all = L_all.cpu().numpy().view(np.uint32)
    return {"point3": P_all.cpu().numpy(), "sem": (lab_all & 0xFFFF).astype(np.uint16),
            "ins": (lab_all >> 16).astype(np.uint16), "counts": C_all.cpu().numpy().astype(np.int64),
            "total": int(P_all.shape[0])}

"""MI355X engine behind the reference's engine interface.

Replaces raycast_engine/raycast_engine_cpu.py:17-111 and raycast_engine/raycast_engine_gpu_simple.py:12-98
(both are Open3D/Embree CPU code in the reference).  All arithmetic runs in the HIP kernels of
liblidarcast; this file only validates arguments, caches scenes and reshapes outputs.  There is no
CPU path: without the built library or without a GPU the constructor raises, which the reference's
caller turns into its own fallback decision (s3dis_simulator.py:66-74).
"""
import hashlib
import itertools
import os
import weakref

import numpy as np

from lidarcast import Context, Scene

try:
    from .raycast_engine import RaycastEngineBase
except ImportError:                      # imported as a top-level package, as the reference's simulator does
    from raycast_engine.raycast_engine import RaycastEngineBase


def mesh_arrays(mesh):
    """Duck-typed mesh access: anything with ``.vertices`` (V,3) and ``.triangles`` (T,3).

    Optional per-triangle ``.triangle_sem`` / ``.triangle_ins`` label arrays are passed through.
    """
    if not (hasattr(mesh, "vertices") and hasattr(mesh, "triangles")):
        raise TypeError("mesh must expose .vertices and .triangles")
    v = np.asarray(mesh.vertices)
    f = np.asarray(mesh.triangles)
    sem = getattr(mesh, "triangle_sem", None)
    ins = getattr(mesh, "triangle_ins", None)
    return v, f, (None if sem is None else np.asarray(sem)), (None if ins is None else np.asarray(ins))


def _fingerprint(v, f, full=False):
    """Cache key of a mesh's arrays.  ``full=False`` (the default on the per-waypoint call path, where the reference
    calls the engine once per pose with the same mesh object) hashes 4 KB from both ends of each array plus 256 rows
    spread evenly over the whole of it: ~20 microseconds (a 65 k-ray call takes 230), enough to notice a replaced or
    re-generated mesh; an in-place edit of a few rows of a large array can escape it -- after editing a mesh in place
    call ``engine.clear_cache()`` (or construct the engine with ``cache_check="full"``, which hashes every byte:
    milliseconds per call on a million-triangle mesh)."""
    h = hashlib.blake2b(digest_size=16)
    for a in (v, f):
        a = np.ascontiguousarray(a)
        b = a.view(np.uint8).reshape(-1)
        if full or b.size <= (1 << 15):
            h.update(b.tobytes())
        else:
            h.update(b[:4096].tobytes())
            h.update(b[-4096:].tobytes())
            rows = a.reshape(len(a), -1)
            h.update(np.ascontiguousarray(rows[::max(1, len(rows) // 256)]).tobytes())
    return (v.shape, f.shape, str(v.dtype), str(f.dtype), h.hexdigest())


_RAY_POOL = None


# threads of the native draws of the seeded stream: one generates words, the others flag and transform; the pool below
# (numpy trigonometry, pose by pose) needs cores beside them
import os as _os
_DRAW_THREADS = int(_os.environ.get("LRC_DRAW_THREADS", "8"))


def _ray_pool():
    """Threads for the host-side ray generation of the dual-axis sensor (pure numpy work per pose)."""
    global _RAY_POOL
    if _RAY_POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _RAY_POOL = ThreadPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 2) - 1)))
    return _RAY_POOL


def _pose_rays_numpy(l, z, u, out, km):
    phi, theta, k = l.scan_angles_from_draws(z, u)
    if k is not None:
        km[:] = k
    l.rays_from_angles(phi, theta, out)


def _pose_rays_native(l, z, u, out, km):
    """One pose's rays and dropout mask from its draws: the angles and the mask as scan_angles_from_draws forms them, the
    sines and cosines by numpy (nothing else reproduces them bit for bit), then products, rotation and narrowing in ONE
    native pass (lrc_rays_from_trig: the reference's un-fused arithmetic) instead of a dozen numpy passes over temporaries."""
    from lidarcast import _capi
    phi, theta = l.scan_pattern(None)
    if z is not None:
        zz = z.reshape(phi.size, 2)
        phi, theta = phi + zz[:, 0], theta + zz[:, 1]
    if u is not None:
        km[:] = u > l.intrinsics.dropout_probability
    ct, st, cp, sp = np.cos(theta), np.sin(theta), np.cos(phi), np.sin(phi)
    M = np.ascontiguousarray(l.pose, dtype=np.float64)
    _capi.check(_capi.load().lrc_rays_from_trig(ct.ctypes.data, st.ctypes.data, cp.ctypes.data, sp.ctypes.data, phi.size,
                                                M.ctypes.data, out.ctypes.data), "lrc_rays_from_trig")


def dual_axis_rays_batch(lidars, rays, keep):
    """All rays of all poses of a dual-axis trajectory BEFORE the dropout, into ``rays`` (P, n, 6) float32, and the dropout
    masks into ``keep`` (P, n) uint8/bool (pre-set to 1) -- ``rays[i][keep[i]]`` is ``lidars[i].get_rays()`` and the
    generator is left where P calls of get_rays() would leave it (reference: lidar/indoor_lidar.py:257-296).  Host only."""
    k0 = lidars[0].intrinsics
    P, n = len(lidars), rays.shape[1]
    # the random draws are sequential by definition of the seeded stream (pose after pose, angles then dropout);
    # the trigonometry and rotation of a pose depend on nothing else and run on a thread pool beside the next
    # poses' draws (numpy releases the GIL in both); at most 48 poses' angle arrays are alive
    import collections
    from lidarcast import nprandom
    pool, pending = _ray_pool(), collections.deque()
    rng = lidars[0].rng
    same = all(l.intrinsics is k0 or l.intrinsics == k0 for l in lidars) and all(l.rng is rng for l in lidars)
    if same and nprandom.supported(rng) and hasattr(lidars[0], "scan_angles_from_draws"):
        # numpy's legacy stream restated natively (csrc/lrc_nprandom.cpp): the draws of a run of poses in ONE call
        # -- the same doubles, the same generator state afterwards -- while the pool turns the previous run's
        # angles into rays
        nn = 2 * n if k0.angle_noise_std > 0 else 0
        nu = n if k0.dropout_probability > 0 else 0

        native_out = rays.dtype == np.float32 and rays[0].flags.c_contiguous
        one = _pose_rays_native if native_out else _pose_rays_numpy
        run = 32
        for a in range(0, P, run):
            b = min(P, a + run)
            z, u = nprandom.scan_draws(b - a, nn, nu, 0.0, k0.angle_noise_std, rng=rng, threads=_DRAW_THREADS)
            for i in range(a, b):
                pending.append(pool.submit(one, lidars[i], z[i - a] if nn else None, u[i - a] if nu else None,
                                           rays[i], keep[i]))
            while len(pending) > 48:
                pending.popleft().result()
    else:
        for i, l in enumerate(lidars):
            phi, theta, k = l.scan_angles()
            if phi.size != n:
                raise ValueError("dual-axis poses must share one ray count")
            if k is not None:
                keep[i] = k
            pending.append(pool.submit(l.rays_from_angles, phi, theta, rays[i]))
            while len(pending) > 48:
                pending.popleft().result()
    for f in pending:
        f.result()


class RaycastEngineHIP(RaycastEngineBase):
    """HIP (gfx950) engine.  ``RaycastEngineGPU`` is this class."""

    # page-locked host blocks reserved at construction: the frame arrays of one 32 x 2048 x 64-pose trajectory (points 50 MB ->
    # a 64 MB block, labels 8.4 MB -> 16 MB blocks each); () or LRC_PRELOCK=0 reserves nothing
    PRELOCK_BYTES = (64 << 20, 16 << 20, 16 << 20)

    def __init__(self, verbose=False, device=0, max_cached_scenes=4, cache_check="sampled", prelock_bytes=None):
        super().__init__()
        self.verbose = verbose
        self.cache_check = cache_check      # "sampled" | "full": see _fingerprint
        self.ctx = Context(device)          # raises when there is no GPU / no library
        # The first scan of a process used to page-lock its frame buffers inside the call (28-30 ms for a C3 trajectory:
        # the whole of C5's first scene).  The blocks are locked here instead, once per engine.  (On a helper thread it
        # raced with the caller's own first HIP calls -- torch's device initialisation failed with "No HIP GPUs are
        # available" in one run -- and the runtime serialises host allocations with other HIP calls anyway.)
        sizes = self.PRELOCK_BYTES if prelock_bytes is None else tuple(prelock_bytes)
        if sizes and os.environ.get("LRC_PRELOCK", "1") != "0":
            self.ctx.pinned.reserve(sizes)
        self._scenes = {}                   # id(mesh) -> (weakref or None, fingerprint, Scene)
        self._dir_tables = {}
        self._grids = {}
        self._dev_tables = {}
        # The packet kernel (lrc_scan_grid_*, csrc/lrc_sector.h) returns the same bytes as the per-ray kernel and is
        # kept as a measured alternative: on the benchmark scenes it is 3-10x SLOWER, so it is
        # off unless asked for.
        self.packet_kernel = False
        self.min_packets = 1024             # with packet_kernel: smaller scans stay on the per-ray kernel anyway
        self._max_cached = int(max_cached_scenes)

    # ---- scene cache: build once per mesh ---------------------------------------------------------
    def scene_for(self, mesh):
        v, f, sem, ins = mesh_arrays(mesh)
        key = id(mesh)
        fp = _fingerprint(v, f, full=self.cache_check == "full")
        ent = self._scenes.get(key)
        if ent is not None and ent[1] == fp and (ent[0] is None or ent[0]() is mesh):
            return ent[2]
        scene = Scene(self.ctx, v, f, sem, ins)
        if self.verbose:
            i = scene.info
            print(f"[lidarcast] scene built: T={i['num_triangles']} nodes={i['num_nodes']} "
                  f"depth={i['max_depth']} build={i['build_ms']:.1f} ms upload={i['upload_ms']:.1f} ms")
        try:
            ref = weakref.ref(mesh)
        except TypeError:
            ref = None
        if len(self._scenes) >= self._max_cached:
            oldest = self._scenes.pop(next(iter(self._scenes)))
            oldest[2].close()
        self._scenes[key] = (ref, fp, scene)
        return scene

    def clear_cache(self):
        for ent in self._scenes.values():
            ent[2].close()
        self._scenes.clear()
        for ent in self._dev_tables.values():
            ent[1].close()
        self._dev_tables.clear()

    # ---- reference interface ------------------------------------------------------------------------
    @staticmethod
    def _check_rays(rays):
        if not isinstance(rays, np.ndarray):
            raise TypeError("rays must be a numpy array.")
        if rays.ndim != 2 or rays.shape[1] != 6:
            raise ValueError("rays must be a (N, 6) array.")

    def cast_rays(self, rays, mesh, center=None, max_range=np.inf, want=None):
        """Per-ray result dict (superset of Open3D's cast_rays dict): t_hit, primitive_ids,
        primitive_normals, points, semantic, instance, incident_angles."""
        self._check_rays(rays)
        scene = self.scene_for(mesh)
        want = want or ("t", "prim", "normal3", "point3", "sem", "ins", "incident_deg")
        out = scene.cast(rays.astype(np.float32), center=center, max_range=max_range, want=want)
        names = {"t": "t_hit", "prim": "primitive_ids", "normal3": "primitive_normals",
                 "point3": "points", "sem": "semantic", "ins": "instance",
                 "incident_deg": "incident_angles"}
        return {names[k]: a for k, a in out.items()}

    def rays_intersect_mesh(self, rays: np.ndarray, mesh):
        self._check_rays(rays)
        scene = self.scene_for(mesh)
        out = scene.cast(rays.astype(np.float32), want=("t", "point3"))
        return out["point3"][out["t"] != np.inf]

    def _direction_table(self, intrinsics):
        """Pose-independent float64 direction table of a multi-line sensor, cached per (elevations, width)."""
        from lidar import IndoorLidar
        k = intrinsics
        if k.vertical_degrees is None:      # uniform-elevation branch: the reference narrows the table to float32 first
            key = ("uniform", float(k.fov_up), float(k.fov_down), int(k.vertical_res), int(k.horizontal_res))
        else:
            key = (tuple(k.vertical_degrees), int(k.horizontal_res))
        tab = self._dir_tables.get(key)
        if tab is None:
            if k.vertical_degrees is None:
                tab = IndoorLidar.directions_uniform(k.fov_up, k.fov_down, k.vertical_res,
                                                     k.horizontal_res).astype(np.float64)
            else:
                tab = IndoorLidar.directions_from_vertical_degrees(k.vertical_degrees, k.horizontal_res)
            if len(self._dir_tables) >= 8:
                self._dir_tables.pop(next(iter(self._dir_tables)))
            self._dir_tables[key] = tab
        return tab

    def _grid_of(self, intrinsics, num_poses=1):
        """(lines, width, az0, az_step) when the sensor's direction table is a (scan line x azimuth) grid the packet
        kernel can take and the scan is large enough to fill the GPU with packets, else None (per-ray kernel).
        The structure is derived from the table and verified entry by entry, once per table."""
        if not self.packet_kernel or getattr(intrinsics, "vertical_degrees", None) is None:
            return None
        tab = self._direction_table(intrinsics)
        key = id(tab)
        ent = self._grids.get(key)
        if ent is None or ent[0] is not tab:
            ent = (tab, self._derive_grid(tab, int(intrinsics.horizontal_res)))
            if len(self._grids) >= 8:
                self._grids.pop(next(iter(self._grids)))
            self._grids[key] = ent
        grid = ent[1]
        if grid is None:
            return None
        packets = int(num_poses) * ((grid[0] + 7) // 8) * (grid[1] // 64)
        return grid if packets >= self.min_packets else None

    @staticmethod
    def _derive_grid(tab, W):
        W = max(1, int(W))
        if W % 64 or W < 256 or len(tab) % W:
            return None
        H = len(tab) // W
        t = tab.reshape(H, W, 3)
        if not np.isfinite(t).all() or not (t[:, :, 2] == t[:, :1, 2]).all():
            return None                                        # one elevation per line, exactly
        ch = np.sqrt(t[:, 0, 0] ** 2 + t[:, 0, 1] ** 2)
        if (ch < 1e-6).any():
            return None                                        # a line looking straight up / down has no azimuth
        az = np.arctan2(t[0, :, 1], t[0, :, 0])
        step = np.angle(np.exp(1j * (az[1] - az[0])))
        if abs(abs(step) * W - 2 * np.pi) > 1e-9:
            return None
        step = float(np.sign(step)) * 2 * np.pi / W            # one turn exactly
        model = az[0] + step * np.arange(W)
        ok = np.abs(t[:, :, 0] - ch[:, None] * np.cos(model)[None, :]).max() < 1e-12 and \
            np.abs(t[:, :, 1] - ch[:, None] * np.sin(model)[None, :]).max() < 1e-12 and \
            np.abs(ch ** 2 + t[:, 0, 2] ** 2 - 1).max() < 1e-12
        return (H, W, float(az[0]), float(step)) if ok else None

    def _resident_table(self, intrinsics):
        """The sensor's direction table as a handle resident in HBM, created on first use, kept with the host table."""
        from lidarcast import DirectionTable
        tab = self._direction_table(intrinsics)
        ent = self._dev_tables.get(id(tab))
        if ent is None or ent[0] is not tab:
            ent = (tab, DirectionTable(self.ctx, tab))
            if len(self._dev_tables) >= 8:
                self._dev_tables.pop(next(iter(self._dev_tables)))[1].close()
            self._dev_tables[id(tab)] = ent
        return ent[1]

    def lidar_intersect_mesh(self, lidar, mesh):
        from lidar import IndoorLidar
        scene = self.scene_for(mesh)
        if type(lidar) is IndoorLidar and (lidar.intrinsics.vertical_degrees is None or len(lidar.intrinsics.vertical_degrees)):
            # this package's own multi-line sensor: rays are generated in the kernel from the direction table
            # (bit-identical to lidar.get_rays(), rotated poses included) instead of on the host
            # ... and compacted in HBM: only the kept points and angles cross PCIe (lrc_scan_poses_compact)
            fr = scene.scan_poses_compact(np.asarray(lidar.pose, dtype=np.float64)[None],
                                          self._resident_table(lidar.intrinsics), lidar.intrinsics.max_range,
                                          want=("point3", "incident_deg"))
            if fr["total"] > 0:
                return fr["point3"], fr["incident_deg"]
            return np.empty((0, 3), dtype=np.float32), np.empty(0)
        else:
            rays = lidar.get_rays()
            self._check_rays(rays)
            out = scene.cast(rays.astype(np.float32), center=np.asarray(lidar.pose)[:3, 3],
                             max_range=lidar.intrinsics.max_range, want=("t", "point3", "incident_deg"))
        keep = out["t"] != np.inf
        points = out["point3"][keep]
        if len(points) > 0:
            return points, out["incident_deg"][keep]
        return points, np.empty(0)

    # ---- pose-batched fast path (what the per-waypoint loop becomes) ---------------------------------
    def scan_poses(self, intrinsics, poses, mesh, want=("t", "point3", "incident_deg")):
        """All poses of a trajectory in one launch, rays generated in the kernel.

        Returns (records dict of (P, N, ...) arrays, N).  Only for sensors with a pose-independent
        direction table (IndoorLidar with vertical_degrees); other sensors go pose by pose.
        """
        from lidar import IndoorLidar
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4, 4)
        if not hasattr(intrinsics, "horizontal_res") or hasattr(intrinsics, "swing_amplitude"):
            raise ValueError("sensor has no pose-independent direction table")
        dirs = self._direction_table(intrinsics)
        scene = self.scene_for(mesh)
        out = scene.scan_poses(poses, dirs, intrinsics.max_range, want=want)
        P, N = poses.shape[0], dirs.shape[0]
        return {k: a.reshape((P, N) + a.shape[1:]) for k, a in out.items()}, N


    def scan_frames(self, intrinsics, poses, mesh, want=("point3", "sem", "ins")):
        """The whole trajectory straight to the reference's per-pose frames: scan + compaction stay in HBM, only the
        kept rows cross PCIe, into page-locked buffers (lrc_scan_poses_compact).  Returns the dict of
        lidarcast.Scene.scan_poses_compact: (K, ...) arrays in np.vstack order + ``counts`` (P,) + ``total``;
        ``split_frames`` turns it into per-pose views.  What the per-waypoint loop of
        s3dis_simulator.py:254-288 produces, for multi-line sensors."""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4, 4)
        if not hasattr(intrinsics, "horizontal_res") or hasattr(intrinsics, "swing_amplitude"):
            raise ValueError("sensor has no pose-independent direction table")
        return self.scene_for(mesh).scan_poses_compact(poses, self._resident_table(intrinsics),
                                                       intrinsics.max_range, want=want,
                                                       grid=self._grid_of(intrinsics, len(poses)))

    def scan_frames_lidars(self, lidars, mesh, want=("point3", "sem", "ins")):
        """The bit-exact default path of the dual-axis sensor, straight to frames: every pose's rays come from the host
        generator (``all_rays_and_mask``: the reference's arithmetic and RNG draws, written into a page-locked buffer),
        ALL of them are uploaded at a fixed stride with the dropout mask, dropped rays are never cast, compaction and
        the per-pose statistics happen in HBM (lrc_scan_rays_compact).  Same frames as ``scan_lidars`` + host masking;
        same return value as ``scan_frames``."""
        if len(lidars) == 0:
            raise ValueError("no lidars given")
        k0 = lidars[0].intrinsics
        n = k0.num_vertical_lines * (int(k0.point_rate * k0.scan_duration) // k0.num_vertical_lines)
        P = len(lidars)
        rays = self.ctx.pinned.take(P * n * 24)[:P * n * 24].view(np.float32).reshape(P, n, 6)
        keep = self.ctx.pinned.take(P * n)[:P * n].reshape(P, n)
        keep[:] = 1
        dual_axis_rays_batch(lidars, rays, keep)
        centers = np.stack([np.asarray(l.pose, dtype=np.float64)[:3, 3] for l in lidars])
        return self.scene_for(mesh).scan_rays_compact(rays, keep, centers, k0.max_range, want=want)

    def scan_frames_dual_axis(self, lidars, mesh, want=("point3", "sem", "ins")):
        """Opt-in fast path of the dual-axis sensor: the noisy scan angles and the dropout mask are drawn on the host
        from the numpy stream, pose after pose, exactly as ``get_rays()`` would draw them (the seeded stream is the
        reference's definition of the scan, lidar/indoor_lidar.py:262-294); trigonometry, rotation and the float32
        narrowing happen in the kernel (lrc_scan_angles_compact).  Not bit-guaranteed against ``scan_lidars`` (device
        vs host sin/cos); same return value as ``scan_frames``."""
        if len(lidars) == 0:
            raise ValueError("no lidars given")
        k0 = lidars[0].intrinsics
        n = k0.num_vertical_lines * (int(k0.point_rate * k0.scan_duration) // k0.num_vertical_lines)
        ang = self.ctx.pinned.take(len(lidars) * n * 16)[:len(lidars) * n * 16].view(np.float64).reshape(len(lidars), n, 2)
        keep = self.ctx.pinned.take(len(lidars) * n)[:len(lidars) * n].reshape(len(lidars), n)
        keep[:] = 1
        for i, l in enumerate(lidars):
            phi, theta, k = l.scan_angles()
            if phi.size != n:
                raise ValueError("dual-axis poses must share one ray count")
            ang[i, :, 0], ang[i, :, 1] = phi, theta
            if k is not None:
                keep[i] = k
        poses = np.stack([np.asarray(l.pose, dtype=np.float64) for l in lidars])
        return self.scene_for(mesh).scan_angles_compact(poses, ang, keep, lidars[0].intrinsics.max_range, want=want)

    # ---- device steps of the N-rank scan (lidarcast.distributed.scan_frames_sharded) ----------------------
    def torch_device(self):
        import torch
        return torch.device("cuda", self.ctx.device)

    def prim_gather(self, poses_local, rays_per_pose, dist, group=None):
        """Send / receive slabs of the per-scan all-gather, cached per shape (PrimGather)."""
        from lidarcast.distributed import PrimGather
        key = (int(poses_local), int(rays_per_pose), id(dist), id(group))
        g = getattr(self, "_gathers", {}).get(key)
        if g is None:
            g = PrimGather(poses_local, rays_per_pose, dist, self.torch_device(), group=group)
            self._gathers = {key: g}
        return g

    def scan_block_into(self, g, intrinsics, block_poses, mesh):
        """Trace this rank's pose block; the kernel writes the hit triangle ids and the per-wave keep counts straight
        into the gather's send slab.  A block shorter than the slab leaves invalid ids / zero counts behind."""
        import torch
        from lidarcast import DeviceHits
        dev = self.torch_device()
        scene = self.scene_for(mesh)
        dirs = self._direction_table(intrinsics)
        block_poses = np.ascontiguousarray(block_poses, dtype=np.float64).reshape(-1, 16)
        p_loc, n = len(block_poses), len(dirs)
        g.wait()
        if p_loc < g.poses_local:
            g.slab.fill_(-1)
            if g.tile_count is not None:
                g.tile_count.zero_()
        if p_loc == 0:
            return
        want = ("prim", "tile_count") if g.fused_counts else ("prim",)
        hits = DeviceHits(0, dev, want=())
        hits.struct.prim = g.prim.data_ptr()
        if g.fused_counts:
            hits.struct.tile_count = g.tile_count.data_ptr()
        d_poses = torch.from_numpy(block_poses).to(dev)
        d_dirs = torch.from_numpy(np.ascontiguousarray(dirs)).to(dev)
        scene.scan_poses_dev(d_poses, d_dirs, hits, intrinsics.max_range, torch.cuda.current_stream().cuda_stream,
                             grid=self._grid_of(intrinsics, p_loc))
        torch.cuda.current_stream().synchronize()       # the slab is complete before the collective reads it

    def cloud_from_gather(self, g, intrinsics, padded_poses, mesh):
        """Every rank's gathered ids -> the compacted (x, y, z, label) rows of ALL poses + per-pose counts, rebuilt
        with the scan's own arithmetic (lrc_cloud_from_prims_dev), then ONE transfer of the kept rows."""
        import torch
        dev = self.torch_device()
        scene = self.scene_for(mesh)
        dirs = self._direction_table(intrinsics)
        poses = np.ascontiguousarray(padded_poses, dtype=np.float64).reshape(-1, 16)
        P, n = len(poses), len(dirs)
        g.wait()
        d_poses = torch.from_numpy(poses).to(dev)
        d_dirs = torch.from_numpy(np.ascontiguousarray(dirs)).to(dev)
        rows = torch.empty((P * n, 4), dtype=torch.float32, device=dev)
        counts = torch.zeros(P, dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        scene.cloud_from_prims_dev(d_poses, d_dirs, g.all_prims, rows, counts, g.all_tile_counts,
                                   poses_per_slab=g.poses_local, slab_stride_bytes=g.stride_bytes, stream=st)
        # the ScanQuality range statistics of every pose, on the assembled rows, with numpy's arithmetic (lrc_stats.h)
        rng = torch.empty(P * n, dtype=torch.float32, device=dev)
        mean = torch.empty(P, dtype=torch.float32, device=dev)
        std = torch.empty(P, dtype=torch.float32, device=dev)
        self.ctx.cloud_range_stats_dev(rows, counts, rng, mean, std, stream=st)
        c = counts.cpu().numpy()
        stats = {"range_origin_mean": mean.cpu().numpy(), "range_origin_std": std.cpu().numpy()}
        return rows[:int(c.sum())].cpu().numpy(), c, stats

    @staticmethod
    def split_frames(frames, name):
        """Per-pose views of one attribute of a scan_frames result (no copy); a tuple of names gives one list each."""
        counts = frames["counts"].tolist()
        ends = list(itertools.accumulate(counts))
        if isinstance(name, (tuple, list)):
            return [[frames[n][e - c:e] for c, e in zip(counts, ends)] for n in name]
        a = frames[name]
        return [a[e - c:e] for c, e in zip(counts, ends)]

    def scan_lidars(self, lidars, mesh, want=("t", "point3", "incident_deg")):
        """Several sensor poses whose rays come from the host generator (dual-axis sensor: seeded noise and
        dropout make the ray sets ragged), cast in ONE launch.  Returns (records dict of flat arrays, offsets):
        pose i owns records[offsets[i]:offsets[i+1]].  ``get_rays()`` is called in list order, so a seeded
        global numpy stream is consumed exactly as the reference's per-waypoint loop would."""
        if len(lidars) == 0:
            raise ValueError("no lidars given")
        rays = [l.get_rays() for l in lidars]
        for r in rays:
            self._check_rays(r)
        off = np.zeros(len(rays) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in rays])
        centers = np.stack([np.asarray(l.pose)[:3, 3] for l in lidars])
        max_range = lidars[0].intrinsics.max_range
        scene = self.scene_for(mesh)
        out = scene.cast_segments(np.concatenate(rays).astype(np.float32), off, centers, max_range, want=want)
        return out, off.astype(np.int64)


class RaycastEngineGPU(RaycastEngineHIP):
    """Drop-in for the reference's RaycastEngineGPU (raycast_engine_gpu_simple.py:12-19)."""

    def __init__(self, verbose=False, **kw):
        super().__init__(verbose=verbose, **kw)


class RaycastEngineCPU(RaycastEngineHIP):
    """Import-compatible name for the reference's RaycastEngineCPU (raycast_engine_cpu.py:17-22).

    This build has no CPU compute path: the class computes on the MI355X exactly like
    RaycastEngineGPU and raises if there is no GPU.
    """

    def __init__(self):
        super().__init__(verbose=False)

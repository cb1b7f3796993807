"""MI355X engine behind the reference's engine interface.

Replaces raycast_engine/raycast_engine_cpu.py:17-111 and raycast_engine/raycast_engine_gpu_simple.py:12-98
(both are Open3D/Embree CPU code in the reference).  All arithmetic runs in the HIP kernels of
liblidarcast; this file only validates arguments, caches scenes and reshapes outputs.  There is no
CPU path: without the built library or without a GPU the constructor raises, which the reference's
caller turns into its own fallback decision (s3dis_simulator.py:66-74).
"""
import hashlib
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


def _fingerprint(v, f):
    h = hashlib.blake2b(digest_size=16)
    h.update(np.ascontiguousarray(v).view(np.uint8).reshape(-1)[:1 << 16].tobytes())
    h.update(np.ascontiguousarray(f).view(np.uint8).reshape(-1)[:1 << 16].tobytes())
    h.update(np.ascontiguousarray(v).view(np.uint8).reshape(-1)[-(1 << 16):].tobytes())
    return (v.shape, f.shape, str(v.dtype), str(f.dtype), h.hexdigest())


class RaycastEngineHIP(RaycastEngineBase):
    """HIP (gfx950) engine.  ``RaycastEngineGPU`` is this class."""

    def __init__(self, verbose=False, device=0, max_cached_scenes=4):
        super().__init__()
        self.verbose = verbose
        self.ctx = Context(device)          # raises when there is no GPU / no library
        self._scenes = {}                   # id(mesh) -> (weakref or None, fingerprint, Scene)
        self._dir_tables = {}
        self._max_cached = int(max_cached_scenes)

    # ---- scene cache: build once per mesh ---------------------------------------------------------
    def scene_for(self, mesh):
        v, f, sem, ins = mesh_arrays(mesh)
        key = id(mesh)
        fp = _fingerprint(v, f)
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

    def lidar_intersect_mesh(self, lidar, mesh):
        from lidar import IndoorLidar
        scene = self.scene_for(mesh)
        if type(lidar) is IndoorLidar and (lidar.intrinsics.vertical_degrees is None or len(lidar.intrinsics.vertical_degrees)):
            # this package's own multi-line sensor: rays are generated in the kernel from the direction table
            # (bit-identical to lidar.get_rays(), rotated poses included) instead of on the host
            out = scene.scan_poses(np.asarray(lidar.pose, dtype=np.float64)[None], self._direction_table(lidar.intrinsics),
                                   lidar.intrinsics.max_range, want=("t", "point3", "incident_deg"))
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

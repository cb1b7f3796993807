"""Context / Scene objects over the C ABI.

Host-array methods (numpy in, numpy out) go through lrc_cast / lrc_scan_poses / lrc_compact.
Device methods (torch CUDA tensors; torch is only the allocator and the stream) go through the
*_dev entry points and never leave HBM.

Reference interface being served: raycast_engine/raycast_engine.py:16-61 (engine ABC) and the
Open3D calls at raycast_engine/raycast_engine_cpu.py:46-53.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (LRC_STATS_WORDS, LrcCompactIO, LrcFrames, LrcGrid, LrcHits, LrcScanOptions, LrcSceneInfo,
                    check)

ATTRS = ("t", "prim", "normal3", "point3", "sem", "ins", "incident_deg")
_NP_SPEC = {
    "t": (np.float32, ()), "prim": (np.uint32, ()), "normal3": (np.float32, (3,)),
    "point3": (np.float32, (3,)), "sem": (np.uint16, ()), "ins": (np.uint16, ()),
    "incident_deg": (np.float64, ()), "intensity": (np.float32, ()),
}


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


FRAME_ATTRS = ("point3", "sem", "ins", "incident_deg", "index", "xyzl", "range_origin")
_FRAME_SPEC = {"point3": (np.float32, (3,)), "sem": (np.uint16, ()), "ins": (np.uint16, ()),
               "incident_deg": (np.float64, ()), "index": (np.uint32, ()), "xyzl": (np.float32, (4,)),
               "range_origin": (np.float32, ())}


class _PinnedBlock:
    """One page-locked host allocation (lrc_host_alloc).  numpy arrays handed out are views of ``buf``; when the
    last view dies the block goes back to its pool (or is freed when the pool is gone or full)."""

    def __init__(self, pool, ptr, nbytes):
        self.pool, self.ptr, self.nbytes = pool, ptr, nbytes


class PinnedPool:
    """Page-locked host buffers for the frame arrays of the *_compact entry points, recycled by size class.

    ``take(nbytes)`` returns a uint8 numpy array over pinned memory.  Arrays sliced / viewed from it keep the
    allocation alive; once the caller drops every view the block returns to the pool, so a simulator that scans
    trajectory after trajectory re-uses the same pinned pages instead of paying hipHostMalloc (milliseconds per
    100 MB) each time, and frames a caller still holds are never overwritten.
    """

    def __init__(self, ctx, max_free_bytes=1 << 30, max_outstanding_bytes=4 << 30):
        self._ctx = ctx
        self._free = {}            # size class -> [ptr, ...]
        self._free_bytes = 0
        self._max_free = int(max_free_bytes)
        self._max_out = int(max_outstanding_bytes)   # page-locked bytes callers may hold at once; beyond it take()
        self.outstanding = 0                          # hands out ordinary (pageable) memory instead of locking more
        self.allocations = 0       # hipHostMalloc calls so far (tests / bench bookkeeping)
        # take() runs on the caller's thread, _release() on whichever thread drops the last view of a block (a pool
        # worker holding frame slices, the garbage collector): the free lists and counters are guarded
        # (re-entrant: dropping the last reference to a slab inside a guarded section runs _release on the spot)
        import threading
        self._lock = threading.RLock()

    @staticmethod
    def _klass(nbytes):
        k = 1 << 16
        while k < nbytes:
            k <<= 1
        return k

    SLAB_BYTES = 32 << 20        # requests up to SMALL_BYTES are carved out of shared slabs of this size
    SMALL_BYTES = 2 << 20        # a single pose's frame arrays (a 65 536-ray pose: 0.8 MB of points); larger requests
                                 # -- a trajectory's arrays -- get blocks of their own, which ARE recycled one by one
    _slab = None                 # [raw ctypes array, fill] of the slab currently being carved

    def _new_block(self, k):
        """A page-locked block of k bytes: from the free list, else hipHostMalloc.  None when the cap is reached."""
        with self._lock:
            lst = self._free.get(k)
            if not lst and self.outstanding + k > self._max_out:
                return None
            self.outstanding += k
            ptr = None
            if lst:
                ptr = lst.pop()
                self._free_bytes -= k
        if ptr is None:
            p = C.c_void_p()
            try:
                check(self._ctx._lib.lrc_host_alloc(self._ctx._h, k, C.byref(p)), "lrc_host_alloc")
            except Exception:
                with self._lock:
                    self.outstanding -= k
                raise
            ptr = p.value
            with self._lock:
                self.allocations += 1
        return ptr

    def _wrap_slab(self, ptr):
        import weakref
        raw = (C.c_uint8 * self.SLAB_BYTES).from_address(ptr)
        weakref.finalize(raw, PinnedPool._release, weakref.ref(self), ptr, self.SLAB_BYTES, self._ctx)
        return raw

    def _take_small(self, nbytes):
        """A slice of a page-locked slab (bump allocation, 256-byte aligned).  A per-waypoint caller -- the reference's own
        loop keeps every frame it is handed -- would otherwise pay a hipHostMalloc per call; this way it pays one per
        32 MB.  A slab returns to the pool when the last array cut from it is dropped.  (Locking the next slab ahead of
        time on a helper thread was tried: the runtime serialises it with the caller's own HIP calls, nothing is gained.)"""
        n = (int(nbytes) + 255) & ~255
        with self._lock:
            sl = self._slab
            if sl is not None and sl[1] + n <= self.SLAB_BYTES:
                off = sl[1]
                sl[1] += n
                return np.frombuffer(sl[0], dtype=np.uint8, count=max(int(nbytes), 1), offset=off)
            self._slab = None          # the old slab lives on through the arrays cut from it
        ptr = self._new_block(self.SLAB_BYTES)
        if ptr is None:
            return np.empty(max(int(nbytes), 1), dtype=np.uint8)
        raw = self._wrap_slab(ptr)
        with self._lock:
            self._slab = [raw, n]
        return np.frombuffer(raw, dtype=np.uint8, count=max(int(nbytes), 1), offset=0)

    def take(self, nbytes):
        import weakref
        if int(nbytes) <= self.SMALL_BYTES:
            return self._take_small(nbytes)
        k = self._klass(max(int(nbytes), 1))
        with self._lock:
            lst = self._free.get(k)
            if not lst and self.outstanding + k > self._max_out:
                return np.empty(k, dtype=np.uint8)      # a caller hoarding frames: stop locking pages, stay correct
            self.outstanding += k
            ptr = None
            if lst:
                ptr = lst.pop()
                self._free_bytes -= k
        if ptr is None:
            p = C.c_void_p()
            try:
                check(self._ctx._lib.lrc_host_alloc(self._ctx._h, k, C.byref(p)), "lrc_host_alloc")
            except Exception:
                with self._lock:
                    self.outstanding -= k
                raise
            ptr = p.value
            with self._lock:
                self.allocations += 1
        raw = (C.c_uint8 * k).from_address(ptr)
        weakref.finalize(raw, PinnedPool._release, weakref.ref(self), ptr, k, self._ctx)
        return np.frombuffer(raw, dtype=np.uint8, count=k)

    def reserve(self, sizes):
        """Page-lock blocks of these sizes NOW and put them on the free lists, so that the first trajectory's frame arrays do
        not pay hipHostMalloc (28-30 ms for the 67 MB of a C3 trajectory) inside the caller's first scan.  Called once, at engine construction."""
        for nbytes in sizes:
            k = self._klass(max(int(nbytes), 1))
            with self._lock:
                if self._free_bytes + k > self._max_free:
                    return
            p = C.c_void_p()
            try:
                check(self._ctx._lib.lrc_host_alloc(self._ctx._h, k, C.byref(p)), "lrc_host_alloc")
            except Exception:
                return
            with self._lock:
                self.allocations += 1
                self._free.setdefault(k, []).append(p.value)
                self._free_bytes += k

    @staticmethod
    def _release(pool_ref, ptr, k, ctx):
        pool = pool_ref()
        keep = False
        if pool is not None:
            with pool._lock:
                pool.outstanding -= k
                if pool._free_bytes + k <= pool._max_free and getattr(ctx, "_h", None):
                    pool._free.setdefault(k, []).append(ptr)
                    pool._free_bytes += k
                    keep = True
        if not keep and getattr(ctx, "_h", None):
            ctx._lib.lrc_host_free(ctx._h, C.c_void_p(ptr))

    def clear(self):
        with self._lock:
            self._slab = None
            lists = list(self._free.values())
            self._free = {}
            self._free_bytes = 0
        for lst in lists:
            for ptr in lst:
                if getattr(self._ctx, "_h", None):
                    self._ctx._lib.lrc_host_free(self._ctx._h, C.c_void_p(ptr))


class Context:
    """One HIP device.  Raises (LidarcastError) when there is no GPU or the library is missing."""

    def __init__(self, device=0):
        self._lib = _capi.load()
        h = C.c_void_p()
        check(self._lib.lrc_ctx_create(int(device), C.byref(h)), "lrc_ctx_create")
        self._h = h
        self.device = int(device)
        self.pinned = PinnedPool(self)

    def synchronize(self):
        check(self._lib.lrc_ctx_synchronize(self._h), "lrc_ctx_synchronize")

    def set_launch_chaining(self, enabled=True):
        """Scans a caller keeps in flight on two streams start one behind the other's last workgroup (include/lidarcast.h)."""
        check(self._lib.lrc_ctx_set_launch_chaining(self._h, 1 if enabled else 0), "lrc_ctx_set_launch_chaining")

    def launch_chaining(self):
        """(enabled, supported by this device)"""
        e, s = C.c_int(0), C.c_int(0)
        check(self._lib.lrc_ctx_get_launch_chaining(self._h, C.byref(e), C.byref(s)), "lrc_ctx_get_launch_chaining")
        return bool(e.value), bool(s.value)

    def close(self):
        if getattr(self, "_h", None):
            self.pinned.clear()
            self._lib.lrc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- compaction -----------------------------------------------------------------------------
    def compact(self, t, seg_len, point3=None, sem=None, ins=None, incident_deg=None,
                want_index=False, want_xyzl=False):
        """Stable compaction of finite-t entries; returns dict(counts, point3, sem, ins, ...)."""
        t = np.ascontiguousarray(t, dtype=np.float32).reshape(-1)
        seg_len = int(seg_len)
        n = t.size
        if seg_len <= 0 or n % seg_len:
            raise ValueError("t.size must be a multiple of seg_len")
        nseg = n // seg_len
        io = LrcCompactIO()
        keep = [t]
        io.t = t.ctypes.data
        counts = np.zeros(nseg, dtype=np.uint64)
        io.counts = counts.ctypes.data
        outs = {}

        def wire(name, arr, dtype, tail):
            if arr is None:
                return
            a = np.ascontiguousarray(arr, dtype=dtype).reshape((n,) + tail)
            keep.append(a)
            setattr(io, name, a.ctypes.data)
            o = np.empty((n,) + tail, dtype=dtype)
            outs[name] = o
            setattr(io, "out_" + name, o.ctypes.data)

        wire("point3", point3, np.float32, (3,))
        wire("sem", sem, np.uint16, ())
        wire("ins", ins, np.uint16, ())
        wire("incident_deg", incident_deg, np.float64, ())
        if want_index:
            outs["index"] = np.empty(n, dtype=np.uint32)
            io.out_index = outs["index"].ctypes.data
        if want_xyzl:
            outs["xyzl"] = np.empty((n, 4), dtype=np.float32)
            io.out_xyzl = outs["xyzl"].ctypes.data
        total = C.c_uint64(0)
        check(self._lib.lrc_compact(self._h, nseg, seg_len, C.byref(io), C.byref(total)), "lrc_compact")
        k = int(total.value)
        res = {"counts": counts.astype(np.int64), "total": k}
        for name, o in outs.items():
            res[name] = o[:k]
        return res

    def cloud_from_ranges_dev(self, poses_t, dirs_t, t_label_t, out_rows_t, counts_t=None, stream=0):
        """Rebuild the compacted (x, y, z, label) rows of a pose-batched scan from its (t, label) pairs."""
        check(self._lib.lrc_cloud_from_ranges_dev(
            self._h, C.c_void_p(poses_t.data_ptr()), poses_t.shape[0], C.c_void_p(dirs_t.data_ptr()),
            dirs_t.shape[0], C.c_void_p(t_label_t.data_ptr()), C.c_void_p(out_rows_t.data_ptr()),
            None if counts_t is None else C.c_void_p(counts_t.data_ptr()), C.c_void_p(int(stream))),
            "lrc_cloud_from_ranges_dev")

    def cloud_range_stats_dev(self, rows_t, counts_t, range_t, mean_t, std_t, stream=0):
        """Per-pose mean / std of |row| (float32, numpy's arithmetic) over assembled (x, y, z, label) rows in HBM."""
        check(self._lib.lrc_cloud_range_stats_dev(
            self._h, C.c_void_p(rows_t.data_ptr()), C.c_void_p(counts_t.data_ptr()), counts_t.shape[0], rows_t.shape[0],
            C.c_void_p(range_t.data_ptr()), C.c_void_p(mean_t.data_ptr()), C.c_void_p(std_t.data_ptr()),
            C.c_void_p(int(stream))), "lrc_cloud_range_stats_dev")

    def compact_dev(self, nseg, seg_len, io, stream=0):
        """io: LrcCompactIO filled with device pointers."""
        check(self._lib.lrc_compact_dev(self._h, int(nseg), int(seg_len), C.byref(io),
                                        C.c_void_p(int(stream))), "lrc_compact_dev")


class DeviceHits:
    """Fixed-stride per-ray records in HBM (torch tensors), n entries."""

    _TORCH = {"t": "float32", "prim": "int32", "normal3": "float32", "point3": "float32",
              "sem": "int16", "ins": "int16", "incident_deg": "float64", "tile_count": "int32",
              "t_label": "int32", "intensity": "float32"}

    def __init__(self, n, device, want=("t", "prim", "normal3", "point3", "sem", "ins")):
        import torch
        self.n = int(n)
        self.want = tuple(want)
        self.tensors = {}
        for a in self.want:
            shape = (self.n, 3) if a.endswith("3") else (self.n,)
            if a == "tile_count":          # kept rays per aligned run of 64 outputs (feeds compact_dev)
                shape = ((self.n + 63) // 64,)
            if a == "t_label":             # packed {float t, uint32 label}: what the multi-GPU gather moves
                shape = (self.n, 2)
            self.tensors[a] = torch.empty(shape, dtype=getattr(torch, self._TORCH[a]), device=device)
        self.struct = LrcHits()
        for a in self.want:
            setattr(self.struct, a, self.tensors[a].data_ptr())

    def __getitem__(self, k):
        return self.tensors[k]

    def bytes_per_ray(self):
        return sum(t.element_size() * (3 if a.endswith("3") else 1) for a, t in self.tensors.items()
                   if a not in ("tile_count", "t_label"))


class DirectionTable:
    """A sensor's (N,3) float64 direction table resident in HBM (lrc_table_create): per-pose callers upload it once."""

    def __init__(self, ctx, dirs):
        self._lib = _capi.load()
        self.ctx = ctx
        d = np.ascontiguousarray(dirs, dtype=np.float64)
        if d.ndim != 2 or d.shape[1] != 3 or len(d) == 0:
            raise ValueError("dirs must be a non-empty (N, 3) array")
        h = C.c_void_p()
        check(self._lib.lrc_table_create(ctx._h, _ptr(d), len(d), C.byref(h)), "lrc_table_create")
        self._h, self.n = h, len(d)

    def __len__(self):
        return self.n

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):
                self._lib.lrc_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """A triangle mesh and its BVH, resident in HBM.  Built once, cast many times."""

    def __init__(self, ctx, vertices, triangles, tri_sem=None, tri_ins=None):
        self._lib = _capi.load()
        self.ctx = ctx
        v = np.ascontiguousarray(np.asarray(vertices), dtype=np.float32)    # the one f64->f32 narrowing
        f = np.asarray(triangles)
        if v.ndim != 2 or v.shape[1] != 3:
            raise ValueError("vertices must be (V, 3)")
        if f.size == 0:
            f = np.zeros((0, 3), dtype=np.uint32)
        if f.ndim != 2 or f.shape[1] != 3:
            raise ValueError("triangles must be (T, 3)")
        if f.dtype == np.int32 and f.flags.c_contiguous:
            # no copy, no pass over the indices: a negative index reads as >= 2^31, which the build's own range check
            # (index < V, on the device) rejects with the same ValueError
            f = f.view(np.uint32)
        else:
            if f.size and f.dtype.kind != "u" and (f.min() < 0):
                raise ValueError("negative triangle index")
            f = np.ascontiguousarray(f, dtype=np.uint32)
        sem = None if tri_sem is None else np.ascontiguousarray(tri_sem, dtype=np.uint16)
        ins = None if tri_ins is None else np.ascontiguousarray(tri_ins, dtype=np.uint16)
        for lab in (sem, ins):
            if lab is not None and lab.shape != (f.shape[0],):
                raise ValueError("per-triangle labels must have shape (T,)")
        h = C.c_void_p()
        check(self._lib.lrc_scene_create(ctx._h, _ptr(v), v.shape[0], _ptr(f), f.shape[0],
                                         _ptr(sem), _ptr(ins), C.byref(h)), "lrc_scene_create")
        self._h = h
        self.num_vertices, self.num_triangles = v.shape[0], f.shape[0]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lrc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def info(self):
        inf = LrcSceneInfo()
        check(self._lib.lrc_scene_get_info(self._h, C.byref(inf)), "lrc_scene_get_info")
        d = {k: getattr(inf, k) for k, _ in LrcSceneInfo._fields_ if not k.startswith("bounds")}
        d["bounds_lo"] = tuple(inf.bounds_lo)
        d["bounds_hi"] = tuple(inf.bounds_hi)
        return d

    def set_options(self, min_range=0.0, range_noise=None, incident_mode=0):
        """Opt-in sensor-realism options (all off = the reference's behaviour); sticky until changed.
        range_noise: float32 array, one entry per ray of the following host-array calls (kept alive here),
        or a (data_ptr, length) pair of a device array for the device calls."""
        o = LrcScanOptions()
        o.min_range = float(min_range)
        o.incident_mode = int(incident_mode)
        self._noise_keepalive = None
        if isinstance(range_noise, tuple):
            o.range_noise, o.range_noise_len = int(range_noise[0]), int(range_noise[1])
        elif range_noise is not None:
            a = np.ascontiguousarray(range_noise, dtype=np.float32).reshape(-1)
            self._noise_keepalive = a
            o.range_noise, o.range_noise_len = a.ctypes.data, a.size
        check(self._lib.lrc_scene_set_options(self._h, C.byref(o)), "lrc_scene_set_options")

    def reset_options(self):
        self._noise_keepalive = None
        check(self._lib.lrc_scene_set_options(self._h, None), "lrc_scene_set_options")

    def occupancy(self):
        """dict(waves_per_cu, vgprs, lds_bytes_per_wave) of the pose-batched trace kernel on this scene."""
        w, v, l = C.c_int(0), C.c_int(0), C.c_int(0)
        check(self._lib.lrc_scene_get_occupancy(self._h, C.byref(w), C.byref(v), C.byref(l)), "lrc_scene_get_occupancy")
        return {"waves_per_cu": w.value, "vgprs": v.value, "lds_bytes_per_wave": l.value}

    def counters(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self._lib.lrc_scene_get_counters(self._h, C.byref(a), C.byref(b)), "lrc_scene_get_counters")
        return int(a.value), int(b.value)

    def export_bvh(self):
        inf = self.info
        nodes = np.zeros((inf["num_nodes"], 16), dtype=np.float32)
        slot_prim = np.zeros(inf["num_slots"], dtype=np.uint32)
        check(self._lib.lrc_scene_export_bvh(self._h, _ptr(nodes), _ptr(slot_prim)), "lrc_scene_export_bvh")
        return nodes, slot_prim

    ARRAYS = {"nodes": 0, "tris": 1, "slot_prim": 2, "slot_label": 3, "prim_plane": 4, "nodes_q": 5, "nodes_n": 6}

    def export_array(self, name):
        """Raw bytes (uint8) of one of the scene's device arrays; empty when the scene has no such array."""
        which = self.ARRAYS[name]
        nbytes = C.c_uint64(0)
        check(self._lib.lrc_scene_export_array(self._h, which, None, 0, C.byref(nbytes)), "lrc_scene_export_array")
        out = np.zeros(int(nbytes.value), dtype=np.uint8)
        if out.size:
            check(self._lib.lrc_scene_export_array(self._h, which, _ptr(out), out.size, None), "lrc_scene_export_array")
        return out

    @classmethod
    def from_device(cls, ctx, verts_t, tris_t, sem_t=None, ins_t=None):
        """Scene from a mesh that is already in HBM (torch CUDA tensors: float32 (V,3), int32/uint32 (T,3), optional
        int16/uint16 (T,) labels): lrc_scene_create_dev, nothing crosses PCIe."""
        self = cls.__new__(cls)
        self._lib = _capi.load()
        self.ctx = ctx
        if verts_t.dim() != 2 or verts_t.shape[1] != 3 or tris_t.dim() != 2 or tris_t.shape[1] != 3:
            raise ValueError("vertices must be (V, 3) and triangles (T, 3)")
        if verts_t.element_size() != 4 or tris_t.element_size() != 4 or not verts_t.is_contiguous() or not tris_t.is_contiguous():
            raise ValueError("device meshes must be contiguous float32 vertices and 32-bit triangle rows")
        for lab in (sem_t, ins_t):
            if lab is not None and (lab.element_size() != 2 or lab.numel() != tris_t.shape[0] or not lab.is_contiguous()):
                raise ValueError("per-triangle labels must be contiguous 16-bit arrays of shape (T,)")
        h = C.c_void_p()
        dp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        check(self._lib.lrc_scene_create_dev(ctx._h, dp(verts_t), verts_t.shape[0], dp(tris_t), tris_t.shape[0],
                                             dp(sem_t), dp(ins_t), C.byref(h)), "lrc_scene_create_dev")
        self._h = h
        self.num_vertices, self.num_triangles = int(verts_t.shape[0]), int(tris_t.shape[0])
        return self

    # ---- host arrays ----------------------------------------------------------------------------
    @staticmethod
    def _alloc(n, want):
        outs, st = {}, LrcHits()
        for a in want:
            if a not in _NP_SPEC:
                raise ValueError(f"unknown hit attribute {a!r}")
            dt, tail = _NP_SPEC[a]
            outs[a] = np.empty((n,) + tail, dtype=dt)
            setattr(st, a, outs[a].ctypes.data)
        return outs, st

    def cast(self, rays, center=None, max_range=np.inf, want=ATTRS):
        """rays (N,6) float32 -> dict of per-ray arrays (Open3D cast_rays + numpy post-processing)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        if rays.ndim != 2 or rays.shape[1] != 6:
            raise ValueError("rays must be a (N, 6) array.")
        n = rays.shape[0]
        outs, st = self._alloc(n, want)
        c = None if center is None else np.ascontiguousarray(center, dtype=np.float64).reshape(3)
        check(self._lib.lrc_cast(self._h, _ptr(rays), n, _ptr(c), float(max_range), C.byref(st)), "lrc_cast")
        return outs

    def cast_segments(self, rays, seg_offsets, centers, max_range, want=ATTRS):
        """Rays of several poses back to back (ragged), one range-filter centre per pose."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        if rays.ndim != 2 or rays.shape[1] != 6:
            raise ValueError("rays must be a (N, 6) array.")
        off = np.ascontiguousarray(seg_offsets, dtype=np.uint64).reshape(-1)
        cen = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, 3)
        if len(off) != len(cen) + 1:
            raise ValueError("seg_offsets must have one more entry than centers")
        n = rays.shape[0]
        outs, st = self._alloc(n, want)
        check(self._lib.lrc_cast_segments(self._h, _ptr(rays), n, _ptr(off), len(cen), _ptr(cen),
                                          float(max_range), C.byref(st)), "lrc_cast_segments")
        return outs

    def scan_poses(self, poses, dirs, max_range, want=ATTRS):
        """poses (P,4,4) f64, dirs (N,3) f64 sensor-frame -> dict of (P*N, ...) arrays."""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 16)
        dirs = np.ascontiguousarray(dirs, dtype=np.float64)
        if dirs.ndim != 2 or dirs.shape[1] != 3:
            raise ValueError("dirs must be (N, 3)")
        P, N = poses.shape[0], dirs.shape[0]
        outs, st = self._alloc(P * N, want)
        check(self._lib.lrc_scan_poses(self._h, _ptr(poses), P, _ptr(dirs), N, float(max_range),
                                       C.byref(st)), "lrc_scan_poses")
        return outs

    # ---- straight to frames ---------------------------------------------------------------------
    PINNED_MIN_BYTES = 0            # every frame array comes from the page-locked pool (small ones are slab slices)

    def _frames_begin(self, P, n, want, capacity):
        cap = int(n if capacity is None else capacity)
        fr = LrcFrames()
        counts = np.zeros(P, dtype=np.uint64)
        fr.counts = counts.ctypes.data
        bufs = {}
        per_pose = {}
        for a in want:
            if a in ("range_origin_stats", "incident_stats"):   # per-pose mean / std with numpy's arithmetic, on the device
                col, dt = ("range_origin", np.float32) if a == "range_origin_stats" else ("incident", np.float64)
                for kind in ("mean", "std"):
                    arr = np.zeros(P, dtype=dt)
                    per_pose[f"{col}_{kind}"] = arr
                    setattr(fr, f"{col}_{kind}", arr.ctypes.data)
                continue
            if a not in _FRAME_SPEC:
                raise ValueError(f"unknown frame attribute {a!r}")
            dt, tail = _FRAME_SPEC[a]
            width = int(np.prod(tail, dtype=np.int64)) if tail else 1
            nbytes = max(cap, 1) * width * np.dtype(dt).itemsize
            if nbytes >= self.PINNED_MIN_BYTES:
                raw = self.ctx.pinned.take(nbytes)
                bufs[a] = raw[:cap * width * np.dtype(dt).itemsize].view(dt).reshape((cap,) + tail)
            else:       # a single pose's worth: page-locking a fresh buffer would cost more than the staged copy saves
                bufs[a] = np.empty((cap,) + tail, dtype=dt)
            setattr(fr, a, bufs[a].ctypes.data)
        bufs["__per_pose__"] = per_pose
        return fr, counts, bufs, cap

    @staticmethod
    def _frames_end(counts, bufs, total):
        per_pose = bufs.pop("__per_pose__")
        out = {a: b[:total] for a, b in bufs.items()}
        out.update(per_pose)
        out["counts"] = counts.astype(np.int64)
        out["total"] = int(total)
        return out

    @staticmethod
    def _grid_struct(grid):
        g = LrcGrid()
        g.lines, g.width, g.az0, g.az_step = int(grid[0]), int(grid[1]), float(grid[2]), float(grid[3])
        return g

    def scan_poses_compact(self, poses, dirs, max_range, want=("point3", "sem", "ins"), capacity=None, grid=None):
        """Pose-batched scan straight to the kept rows of every pose (lrc_scan_poses_compact): dict of (K, ...)
        arrays over page-locked memory + ``counts`` (P,) int64 + ``total`` K.  Frame p = rows
        [counts[:p].sum(), counts[:p+1].sum()).  ``grid`` = (lines, width, az0, az_step) of a table that is a
        (scan line x azimuth) grid selects the packet kernel (lrc_scan_grid_compact): same bytes (a measured alternative,
        slower on the benchmark scenes).  ``dirs`` may be a DirectionTable handle (uploaded once, lrc_scan_table_compact)."""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 16)
        if isinstance(dirs, DirectionTable):          # resident table: nothing to upload
            if not dirs._h:
                raise ValueError("direction table handle is closed")
            if dirs.ctx is not self.ctx:
                raise ValueError("direction table belongs to another context (device)")
            P, N = poses.shape[0], dirs.n
            fr, counts, bufs, cap = self._frames_begin(P, P * N, want, capacity)
            total = C.c_uint64(0)
            g = None if grid is None else self._grid_struct(grid)
            check(self._lib.lrc_scan_table_compact(self._h, _ptr(poses), P, dirs._h, None if g is None else C.byref(g),
                                                   float(max_range), C.byref(fr), cap, C.byref(total)),
                  "lrc_scan_table_compact")
            return self._frames_end(counts, bufs, total.value)
        dirs = np.ascontiguousarray(dirs, dtype=np.float64)
        if dirs.ndim != 2 or dirs.shape[1] != 3:
            raise ValueError("dirs must be (N, 3)")
        P, N = poses.shape[0], dirs.shape[0]
        fr, counts, bufs, cap = self._frames_begin(P, P * N, want, capacity)
        total = C.c_uint64(0)
        if grid is not None:
            g = self._grid_struct(grid)
            check(self._lib.lrc_scan_grid_compact(self._h, _ptr(poses), P, _ptr(dirs), C.byref(g), float(max_range),
                                                  C.byref(fr), cap, C.byref(total)), "lrc_scan_grid_compact")
        else:
            check(self._lib.lrc_scan_poses_compact(self._h, _ptr(poses), P, _ptr(dirs), N, float(max_range),
                                                   C.byref(fr), cap, C.byref(total)), "lrc_scan_poses_compact")
        return self._frames_end(counts, bufs, total.value)

    def scan_angles_compact(self, poses, angles, keep, max_range, want=("point3", "sem", "ins"), capacity=None):
        """Dual-axis sensor, rays generated in the kernel from host-drawn (phi, theta) (lrc_scan_angles_compact).
        angles: (P, N, 2) float64; keep: (P, N) bool / uint8 or None."""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 16)
        angles = np.ascontiguousarray(angles, dtype=np.float64)
        P = poses.shape[0]
        if angles.ndim != 3 or angles.shape[0] != P or angles.shape[2] != 2:
            raise ValueError("angles must be (P, N, 2)")
        N = angles.shape[1]
        k8 = None
        if keep is not None:
            k8 = np.ascontiguousarray(keep).reshape(P, N).view(np.uint8) if np.asarray(keep).dtype == np.bool_ \
                else np.ascontiguousarray(keep, dtype=np.uint8).reshape(P, N)
        fr, counts, bufs, cap = self._frames_begin(P, P * N, want, capacity)
        total = C.c_uint64(0)
        check(self._lib.lrc_scan_angles_compact(self._h, _ptr(poses), P, _ptr(angles), _ptr(k8), N, float(max_range),
                                                C.byref(fr), cap, C.byref(total)), "lrc_scan_angles_compact")
        return self._frames_end(counts, bufs, total.value)

    def scan_rays_compact(self, rays, keep, centers, max_range, want=("point3", "sem", "ins"), capacity=None):
        """Host-generated rays of several poses at a fixed stride straight to frames (lrc_scan_rays_compact).
        rays: (P, N, 6) float32; keep: (P, N) bool / uint8 or None; centers: (P, 3) float64."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        if rays.ndim != 3 or rays.shape[2] != 6:
            raise ValueError("rays must be (P, N, 6)")
        P, N = rays.shape[0], rays.shape[1]
        cen = np.ascontiguousarray(centers, dtype=np.float64).reshape(P, 3)
        k8 = None
        if keep is not None:
            k8 = np.ascontiguousarray(keep).reshape(P, N).view(np.uint8) if np.asarray(keep).dtype == np.bool_ \
                else np.ascontiguousarray(keep, dtype=np.uint8).reshape(P, N)
        fr, counts, bufs, cap = self._frames_begin(P, P * N, want, capacity)
        total = C.c_uint64(0)
        check(self._lib.lrc_scan_rays_compact(self._h, _ptr(rays), _ptr(k8), _ptr(cen), P, N, float(max_range),
                                              C.byref(fr), cap, C.byref(total)), "lrc_scan_rays_compact")
        return self._frames_end(counts, bufs, total.value)

    def scan_stats(self, poses, dirs, max_range):
        """(P*N, 5) uint32 traversal counters per ray from the instrumented trace kernel (lrc_debug_scan_stats):
        node steps, triangle tests, wave-uniform node steps, dead node steps, pad-clause rejections."""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 16)
        dirs = np.ascontiguousarray(dirs, dtype=np.float64)
        P, N = poses.shape[0], dirs.shape[0]
        st = np.zeros((P * N, LRC_STATS_WORDS), dtype=np.uint32)
        check(self._lib.lrc_debug_scan_stats(self._h, _ptr(poses), P, _ptr(dirs), N, float(max_range), _ptr(st)),
              "lrc_debug_scan_stats")
        return st

    # ---- device tensors -------------------------------------------------------------------------
    def scan_angles_dev(self, poses_t, angles_t, keep_t, hits, max_range, stream=0):
        P = poses_t.shape[0]
        N = angles_t.shape[0] // max(P, 1) if angles_t.dim() == 2 else angles_t.shape[1]
        check(self._lib.lrc_scan_angles_dev(self._h, C.c_void_p(poses_t.data_ptr()), P,
                                            C.c_void_p(angles_t.data_ptr()),
                                            None if keep_t is None else C.c_void_p(keep_t.data_ptr()), N,
                                            float(max_range), C.byref(hits.struct), C.c_void_p(int(stream))),
              "lrc_scan_angles_dev")

    def cast_dev(self, rays_t, hits, center=None, max_range=np.inf, stream=0):
        c = None if center is None else np.ascontiguousarray(center, dtype=np.float64).reshape(3)
        check(self._lib.lrc_cast_dev(self._h, C.c_void_p(rays_t.data_ptr()), rays_t.shape[0], _ptr(c),
                                     float(max_range), C.byref(hits.struct), C.c_void_p(int(stream))),
              "lrc_cast_dev")

    def scan_poses_dev(self, poses_t, dirs_t, hits, max_range, stream=0, grid=None):
        P, N = poses_t.shape[0], dirs_t.shape[0]
        if grid is not None:       # the table is a (scan line x azimuth) grid: packet kernel, same bytes
            g = self._grid_struct(grid)
            check(self._lib.lrc_scan_grid_dev(self._h, C.c_void_p(poses_t.data_ptr()), P,
                                              C.c_void_p(dirs_t.data_ptr()), C.byref(g), float(max_range),
                                              C.byref(hits.struct), C.c_void_p(int(stream))), "lrc_scan_grid_dev")
            return
        check(self._lib.lrc_scan_poses_dev(self._h, C.c_void_p(poses_t.data_ptr()), P,
                                           C.c_void_p(dirs_t.data_ptr()), N, float(max_range),
                                           C.byref(hits.struct), C.c_void_p(int(stream))),
              "lrc_scan_poses_dev")

    def cloud_from_prims_dev(self, poses_t, dirs_t, prim_t, out_rows_t, counts_t=None, tile_count_t=None,
                             poses_per_slab=0, slab_stride_bytes=0, stream=0, own_slab=None, own_io=None):
        """Rebuild the compacted (x, y, z, label) rows of a pose-batched scan from its 4-byte triangle ids
        (lrc_hits.prim), in place over the gathered send slabs of several ranks (lrc_cloud_from_prims_dev).
        ``own_slab`` + ``own_io`` (LrcCompactIO with t / point3 / sem / ins of that slab's poses): the caller's own rows
        come from its local records instead of being rebuilt (lrc_cloud_from_prims_own_dev)."""
        if own_slab is not None:
            check(self._lib.lrc_cloud_from_prims_own_dev(
                self._h, C.c_void_p(poses_t.data_ptr()), poses_t.shape[0], C.c_void_p(dirs_t.data_ptr()),
                dirs_t.shape[0], C.c_void_p(prim_t.data_ptr()), C.c_void_p(tile_count_t.data_ptr()),
                int(poses_per_slab), int(slab_stride_bytes), int(own_slab), C.byref(own_io),
                C.c_void_p(out_rows_t.data_ptr()), None if counts_t is None else C.c_void_p(counts_t.data_ptr()),
                C.c_void_p(int(stream))), "lrc_cloud_from_prims_own_dev")
            return
        check(self._lib.lrc_cloud_from_prims_dev(
            self._h, C.c_void_p(poses_t.data_ptr()), poses_t.shape[0], C.c_void_p(dirs_t.data_ptr()),
            dirs_t.shape[0], C.c_void_p(prim_t.data_ptr()),
            None if tile_count_t is None else C.c_void_p(tile_count_t.data_ptr()),
            int(poses_per_slab), int(slab_stride_bytes), C.c_void_p(out_rows_t.data_ptr()),
            None if counts_t is None else C.c_void_p(counts_t.data_ptr()), C.c_void_p(int(stream))),
            "lrc_cloud_from_prims_dev")


class ScanPipe:
    """Consecutive pose batches of one scene, scanned and compacted with the launches overlapped inside the library
    (lrc_pipe_*: the trace of batch k+1 fills the wave slots the trace of batch k leaves empty in its tail; the rows of
    batch k are scattered by the leading workgroups of the trace launch of batch k+2).  Everything stays in HBM; `submit` only enqueues, `wait` orders a stream
    behind all submits so far.  The poses of a trajectory are independent (reference: s3dis_simulator.py:254-288)."""

    def __init__(self, scene, max_poses, rays_per_pose):
        self._lib = _capi.load()
        self.scene = scene
        self.max_poses, self.rays_per_pose = int(max_poses), int(rays_per_pose)
        h = C.c_void_p()
        check(self._lib.lrc_pipe_create(scene._h, self.max_poses, self.rays_per_pose, C.byref(h)), "lrc_pipe_create")
        self._h = h

    def submit(self, poses_t, dirs_t, max_range, io=None, out_rows_t=None, counts_t=None, stream=0):
        """io: LrcCompactIO whose out_* members / counts point at the caller's device buffers; or pass torch tensors
        (out_rows_t: (K,4) float32 rows x, y, z, label bits; counts_t: (P,) int64).  Returns the submit's ticket."""
        if dirs_t.shape[0] != self.rays_per_pose:
            raise ValueError("direction table does not match the pipeline's rays_per_pose")
        if io is None:
            io = _capi.LrcCompactIO()
            if out_rows_t is not None:
                io.out_xyzl = out_rows_t.data_ptr()
            if counts_t is not None:
                io.counts = counts_t.data_ptr()
        ticket = C.c_uint64(0)
        check(self._lib.lrc_pipe_submit(self._h, C.c_void_p(poses_t.data_ptr()), poses_t.shape[0],
                                        C.c_void_p(dirs_t.data_ptr()), float(max_range), C.byref(io),
                                        C.c_void_p(int(stream)), C.byref(ticket)), "lrc_pipe_submit")
        return ticket.value

    def wait(self, stream=0):
        check(self._lib.lrc_pipe_wait(self._h, C.c_void_p(int(stream))), "lrc_pipe_wait")

    # ---- N ranks: ids + keep counts into the caller's send slab, an earlier gathered scan assembled in the launch's front ----
    @staticmethod
    def gathered(all_poses_t, all_prims_t, all_tile_counts_t, poses_per_slab, slab_stride_bytes, own_slab, own_ticket,
                 out_rows_t, counts_t=None, scan_slot=0):
        """An lrc_gathered for submit_sharded / assemble: the slabs of ALL ranks after the collective (torch tensors)."""
        g = _capi.LrcGathered()
        g.d_all_poses16, g.num_poses_all = all_poses_t.data_ptr(), all_poses_t.shape[0]
        g.d_all_prims, g.d_all_tile_counts = all_prims_t.data_ptr(), all_tile_counts_t.data_ptr()
        g.poses_per_slab, g.slab_stride_bytes = int(poses_per_slab), int(slab_stride_bytes)
        g.own_slab, g.own_ticket, g.scan_slot = int(own_slab), int(own_ticket), int(scan_slot)
        g.d_out_xyzl = out_rows_t.data_ptr()
        g.d_counts = counts_t.data_ptr() if counts_t is not None else None
        return g

    def submit_sharded(self, poses_t, dirs_t, max_range, send_prim_t, send_tile_count_t, assemble=None, stream=0):
        ticket = C.c_uint64(0)
        check(self._lib.lrc_pipe_submit_sharded(
            self._h, C.c_void_p(poses_t.data_ptr()), poses_t.shape[0], C.c_void_p(dirs_t.data_ptr()), float(max_range),
            C.c_void_p(send_prim_t.data_ptr()), C.c_void_p(send_tile_count_t.data_ptr()),
            C.byref(assemble) if assemble is not None else None, C.c_void_p(int(stream)), C.byref(ticket)),
            "lrc_pipe_submit_sharded")
        return ticket.value

    def trace_done(self, ticket, stream=0):
        """``stream`` waits for the trace of that submit: its send slab is complete, the collective may start."""
        check(self._lib.lrc_pipe_trace_done(self._h, int(ticket), C.c_void_p(int(stream))), "lrc_pipe_trace_done")

    def scan_gathered(self, dirs_t, gathered, stream=0):
        """The scan over the gathered keep counts: enqueue on the communication stream right behind the collective."""
        check(self._lib.lrc_pipe_scan_gathered(self._h, C.c_void_p(dirs_t.data_ptr()), C.byref(gathered), C.c_void_p(int(stream))),
              "lrc_pipe_scan_gathered")

    def assemble(self, dirs_t, gathered, stream=0):
        check(self._lib.lrc_pipe_assemble(self._h, C.c_void_p(dirs_t.data_ptr()), C.byref(gathered), C.c_void_p(int(stream))),
              "lrc_pipe_assemble")

    def records(self, ticket):
        """LrcHits (device pointers) of the fixed-stride records of that submit; valid until two further submits."""
        h = _capi.LrcHits()
        check(self._lib.lrc_pipe_records(self._h, int(ticket), C.byref(h)), "lrc_pipe_records")
        return h

    def trace_ms(self, ticket):
        ms = C.c_float(0.0)
        check(self._lib.lrc_pipe_trace_ms(self._h, int(ticket), C.byref(ms)), "lrc_pipe_trace_ms")
        return float(ms.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lrc_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NearestIndex:
    """Exact 1-NN into an annotated cloud on the GPU (reference: sklearn ball_tree query,
    containers/s3dis_sim_scene.py:416-418)."""

    def __init__(self, ctx, points, cell_size=0.0):
        self._lib = _capi.load()
        self.ctx = ctx
        p = np.ascontiguousarray(np.asarray(points), dtype=np.float64)
        if p.ndim != 2 or p.shape[1] != 3 or len(p) == 0:
            raise ValueError("points must be a non-empty (M, 3) array")
        h = C.c_void_p()
        check(self._lib.lrc_nn_create(ctx._h, _ptr(p), len(p), float(cell_size), C.byref(h)), "lrc_nn_create")
        self._h = h
        self.num_points = len(p)

    def query(self, points, return_distance=False):
        q = np.ascontiguousarray(np.asarray(points), dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != 3:
            raise ValueError("query points must be (K, 3)")
        idx = np.empty(len(q), dtype=np.uint32)
        dist = np.empty(len(q), dtype=np.float64) if return_distance else None
        check(self._lib.lrc_nn_query(self._h, _ptr(q), len(q), _ptr(idx), _ptr(dist)), "lrc_nn_query")
        return (idx, dist) if return_distance else idx

    def query_dev(self, q_t, idx_t, dist_t=None, stream=0):
        check(self._lib.lrc_nn_query_dev(self._h, C.c_void_p(q_t.data_ptr()), q_t.shape[0],
                                         C.c_void_p(idx_t.data_ptr()),
                                         None if dist_t is None else C.c_void_p(dist_t.data_ptr()),
                                         C.c_void_p(int(stream))), "lrc_nn_query_dev")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lrc_nn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bake_triangle_labels(index, vertices, triangles, semantic, instance):
    """Per-triangle (sem, ins) from the annotated point nearest to each triangle centroid: labels are
    attached once per mesh and written back by the trace kernel, instead of one 1-NN query per hit point at
    export time (reference: containers/s3dis_sim_scene.py:339-377)."""
    v = np.asarray(vertices, dtype=np.float64)
    f = np.asarray(triangles)
    cen = (v[f[:, 0]] + v[f[:, 1]] + v[f[:, 2]]) / 3.0
    nearest = index.query(cen.astype(np.float32))
    return (np.asarray(semantic)[nearest].astype(np.uint16), np.asarray(instance)[nearest].astype(np.uint16))


class OccupancyIndex:
    """Mesh vertices resident in HBM for the planner's robot-cube test (reference:
    trajectory/auto_trajectory_generator.py:219-238)."""

    def __init__(self, ctx, vertices):
        self._lib = _capi.load()
        self.ctx = ctx
        v = np.ascontiguousarray(np.asarray(vertices), dtype=np.float64).reshape(-1, 3)
        h = C.c_void_p()
        check(self._lib.lrc_occ_create(ctx._h, _ptr(v), len(v), C.byref(h)), "lrc_occ_create")
        self._h = h

    def occupied(self, points, half):
        """bool (Q,): does the cube [p - half, p + half] contain a vertex?"""
        p = np.ascontiguousarray(np.asarray(points), dtype=np.float64).reshape(-1, 3)
        flags = np.zeros(len(p), dtype=np.uint8)
        check(self._lib.lrc_occ_query(self._h, _ptr(p), len(p), float(half), _ptr(flags)), "lrc_occ_query")
        return flags.astype(bool)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lrc_occ_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""ctypes loader of oracle/liblrc_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "liblrc_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "lrc_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liblrc_oracle.so"])
    return SO


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(SO)
        vp, u64 = C.c_void_p, C.c_uint64
        lib.orc_cast_brute.argtypes = [vp, vp, u64, vp, u64, vp, vp]
        lib.orc_cast_brute.restype = None
        lib.orc_bvh_build.argtypes = [vp, vp, u64]
        lib.orc_bvh_build.restype = vp
        lib.orc_bvh_free.argtypes = [vp]
        lib.orc_bvh_free.restype = None
        lib.orc_cast_bvh.argtypes = [vp, vp, u64, vp, vp, C.c_int]
        lib.orc_cast_bvh.restype = None
        lib.orc_cast_bvh_diag.argtypes = [vp, vp, u64, vp, vp, vp, C.c_int]
        lib.orc_cast_bvh_diag.restype = None
        lib.orc_witness_f64.argtypes = [vp, vp, u64, vp, vp, vp, C.c_int]
        lib.orc_witness_f64.restype = None
        lib.orc_witness_tri_f64.argtypes = [vp, vp, vp, vp, u64, vp, vp]
        lib.orc_witness_tri_f64.restype = None
        lib.orc_normals.argtypes = [vp, vp, vp, u64, vp]
        lib.orc_normals.restype = None
        lib.orc_has_fma.restype = C.c_int
        _lib = lib
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


class OracleMesh:
    """float32 vertices / uint32 triangles as Open3D's from_legacy would narrow them
    (reference: raycast_engine/raycast_engine_cpu.py:47)."""

    def __init__(self, vertices, triangles):
        self.v = np.ascontiguousarray(np.asarray(vertices), dtype=np.float32)
        f = np.asarray(triangles)
        self.f = np.ascontiguousarray(f.reshape(-1, 3) if f.size else np.zeros((0, 3)), dtype=np.uint32)
        self._bvh = None

    def brute(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = len(rays)
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32)
        load().orc_cast_brute(_p(self.v), _p(self.f), len(self.f), _p(rays), n, _p(t), _p(prim))
        return t, prim

    def build(self):
        self.free()
        self._bvh = load().orc_bvh_build(_p(self.v), _p(self.f), len(self.f))
        if not self._bvh:
            raise MemoryError("orc_bvh_build failed")
        return self

    def cast(self, rays, threads=1):
        if self._bvh is None:
            self.build()
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = len(rays)
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32)
        load().orc_cast_bvh(self._bvh, _p(rays), n, _p(t), _p(prim), int(threads))
        return t, prim

    def cast_diag(self, rays, threads=1):
        """(t, prim, pad_rejections): the cast plus, per ray, how many visited triangles were rejected by the
        padded-box clause alone (the clause Embree does not have)."""
        if self._bvh is None:
            self.build()
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = len(rays)
        t, prim, rej = np.empty(n, np.float32), np.empty(n, np.uint32), np.empty(n, np.uint32)
        load().orc_cast_bvh_diag(self._bvh, _p(rays), n, _p(t), _p(prim), _p(rej), int(threads))
        return t, prim, rej

    def witness(self, rays, threads=1):
        """float64 witness (textbook Moeller-Trumbore in double, no box clause): (t64, prim, margin) with
        margin = min(u, v, 1-u-v) of the winning hit, -1 on a miss."""
        if self._bvh is None:
            self.build()
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = len(rays)
        t, prim, m = np.empty(n, np.float64), np.empty(n, np.uint32), np.empty(n, np.float64)
        load().orc_witness_f64(self._bvh, _p(rays), n, _p(t), _p(prim), _p(m), int(threads))
        return t, prim, m

    def witness_triangle(self, rays, prim):
        """float64 test of the ONE triangle prim[i] per ray: (t64, margin), +inf / -1 where double says miss."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        prim = np.ascontiguousarray(prim, dtype=np.uint32)
        n = len(rays)
        t, m = np.empty(n, np.float64), np.empty(n, np.float64)
        load().orc_witness_tri_f64(_p(self.v), _p(self.f), _p(rays), _p(prim), n, _p(t), _p(m))
        return t, m

    def normals(self, prim):
        prim = np.ascontiguousarray(prim, dtype=np.uint32)
        out = np.empty((len(prim), 3), np.float32)
        load().orc_normals(_p(self.v), _p(self.f), _p(prim), len(prim), _p(out))
        return out

    def free(self):
        if self._bvh is not None:
            load().orc_bvh_free(self._bvh)
            self._bvh = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

"""numpy's legacy seeded stream at native speed (csrc/lrc_nprandom.cpp, lrc_rng_scan_draws).

The reference's dual-axis generator draws from the GLOBAL numpy stream, per pose: two normals per ray, then one
uniform per ray (reference: lidar/indoor_lidar.py:257-296).  ``scan_draws`` returns exactly those doubles for a whole
run of poses and leaves the generator where numpy would have left it, so seeded runs stay bit-identical and whatever
the caller draws afterwards continues the same stream.  Host code; needs no GPU."""
import ctypes as C

import numpy as np

from . import _capi


def _state_of(rng):
    """(get, set) for the legacy generator behind ``rng``: the np.random module (global stream) or a RandomState."""
    if rng is None or rng is np.random:
        return np.random.get_state, np.random.set_state
    if isinstance(rng, np.random.RandomState):
        return rng.get_state, rng.set_state
    return None


def supported(rng):
    """True when ``rng`` is numpy's legacy MT19937 stream (np.random itself or a RandomState over MT19937)."""
    gs = _state_of(rng)
    if gs is None:
        return False
    try:
        return gs[0]()[0] == "MT19937"
    except Exception:
        return False


def scan_draws(num_poses, normals_per_pose, uniforms_per_pose, loc=0.0, scale=1.0, rng=None, threads=0):
    """(normals (P, normals_per_pose), uniforms (P, uniforms_per_pose)) float64: what
    ``[(rng.normal(loc, scale, normals_per_pose), rng.random_sample(uniforms_per_pose)) for _ in range(P)]`` returns,
    with the generator state advanced accordingly.  ``rng``: None / np.random = the global stream, or a RandomState."""
    gs = _state_of(rng)
    if gs is None:
        raise TypeError("scan_draws needs numpy's legacy stream: np.random or a numpy.random.RandomState")
    get_state, set_state = gs
    name, key, pos, has_gauss, gauss = get_state()
    if name != "MT19937":
        raise TypeError(f"legacy generator {name!r} is not MT19937")
    st = _capi.LrcMt19937State()
    C.memmove(st.key, np.ascontiguousarray(key, dtype=np.uint32).ctypes.data, 624 * 4)
    st.pos, st.has_gauss, st.gauss = int(pos), int(has_gauss), float(gauss)
    P, nn, nu = int(num_poses), int(normals_per_pose), int(uniforms_per_pose)
    normals = np.empty((P, nn), dtype=np.float64)
    uniforms = np.empty((P, nu), dtype=np.float64)
    lib = _capi.load()
    _capi.check(lib.lrc_rng_scan_draws(C.byref(st), P, nn, nu, float(loc), float(scale),
                                       C.c_void_p(normals.ctypes.data) if normals.size else None,
                                       C.c_void_p(uniforms.ctypes.data) if uniforms.size else None, int(threads)),
                "lrc_rng_scan_draws")
    set_state((name, np.frombuffer(st.key, dtype=np.uint32).copy(), int(st.pos), int(st.has_gauss), float(st.gauss)))
    return normals, uniforms

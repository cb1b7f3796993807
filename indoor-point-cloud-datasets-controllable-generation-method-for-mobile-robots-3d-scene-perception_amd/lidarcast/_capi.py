"""ctypes binding of liblidarcast (include/lidarcast.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950) and lives next to
this package.  There is no fallback: if the library is missing, or there is no GPU, the calls
raise -- the reference's caller catches engine construction errors itself
(reference: s3dis_simulator.py:66-74).
"""
import ctypes as C
import os
import sys

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("LRC_LIB") or os.path.join(_PKG_ROOT, "liblidarcast.so")   # LRC_LIB: A/B builds

LRC_OK = 0
LRC_ERR_INVALID_ARG = -1
LRC_ERR_NO_DEVICE = -2
LRC_ERR_HIP = -3
LRC_ERR_OOM = -4
LRC_ERR_INTERNAL = -5
LRC_INVALID_PRIM = 0xFFFFFFFF

# every symbol include/lidarcast.h declares (tests check that the library exports all of them)
SYMBOLS = (
    "lrc_version", "lrc_last_error", "lrc_device_count",
    "lrc_ctx_create", "lrc_ctx_destroy", "lrc_ctx_synchronize", "lrc_ctx_set_launch_chaining", "lrc_ctx_get_launch_chaining",
    "lrc_scene_create", "lrc_scene_create_dev", "lrc_scene_destroy", "lrc_scene_get_info", "lrc_scene_export_bvh", "lrc_scene_export_array",
    "lrc_scene_get_counters", "lrc_scene_set_options", "lrc_scene_get_occupancy",
    "lrc_cast", "lrc_cast_dev", "lrc_cast_segments", "lrc_cast_segments_dev",
    "lrc_scan_poses", "lrc_scan_poses_dev", "lrc_scan_poses_compact", "lrc_host_alloc", "lrc_host_free",
    "lrc_pipe_create", "lrc_pipe_destroy", "lrc_pipe_submit", "lrc_pipe_wait", "lrc_pipe_records", "lrc_pipe_trace_ms",
    "lrc_pipe_submit_sharded", "lrc_pipe_trace_done", "lrc_pipe_scan_gathered", "lrc_pipe_assemble",
    "lrc_scan_angles_dev", "lrc_scan_angles_compact", "lrc_debug_scan_stats",
    "lrc_scan_grid_dev", "lrc_scan_grid_compact", "lrc_scan_rays_compact",
    "lrc_table_create", "lrc_table_destroy", "lrc_scan_table_compact",
    "lrc_compact", "lrc_compact_dev", "lrc_cloud_from_ranges_dev", "lrc_cloud_from_prims_dev",
    "lrc_cloud_from_prims_own_dev", "lrc_cloud_range_stats_dev",
    "lrc_nn_create", "lrc_nn_destroy", "lrc_nn_query", "lrc_nn_query_dev",
    "lrc_min_distances", "lrc_rbf_kernel_sum",
    "lrc_occ_create", "lrc_occ_destroy", "lrc_occ_query",
    "lrc_rng_scan_draws", "lrc_rays_from_trig",
)


class LrcHits(C.Structure):
    _fields_ = [("t", C.c_void_p), ("prim", C.c_void_p), ("normal3", C.c_void_p),
                ("point3", C.c_void_p), ("sem", C.c_void_p), ("ins", C.c_void_p),
                ("incident_deg", C.c_void_p), ("t_label", C.c_void_p), ("tile_count", C.c_void_p),
                ("intensity", C.c_void_p)]


class LrcSceneInfo(C.Structure):
    _fields_ = [("num_vertices", C.c_uint64), ("num_triangles", C.c_uint64),
                ("num_nodes", C.c_uint64), ("num_leaves", C.c_uint64), ("num_slots", C.c_uint64),
                ("max_depth", C.c_uint32), ("max_leaf_size", C.c_uint32),
                ("device_bytes", C.c_uint64), ("build_ms", C.c_double), ("upload_ms", C.c_double),
                ("bounds_lo", C.c_float * 3), ("bounds_hi", C.c_float * 3),
                ("quantised_nodes", C.c_uint32), ("leaf_inflation", C.c_float),
                ("device_build", C.c_uint32), ("reserved_", C.c_uint32)]


class LrcScanOptions(C.Structure):
    _fields_ = [("min_range", C.c_double), ("range_noise", C.c_void_p), ("range_noise_len", C.c_uint64),
                ("incident_mode", C.c_int)]


class LrcCompactIO(C.Structure):
    _fields_ = [("t", C.c_void_p), ("point3", C.c_void_p), ("sem", C.c_void_p), ("ins", C.c_void_p),
                ("incident_deg", C.c_void_p), ("tile_count", C.c_void_p), ("counts", C.c_void_p),
                ("out_point3", C.c_void_p), ("out_sem", C.c_void_p), ("out_ins", C.c_void_p),
                ("out_incident_deg", C.c_void_p), ("out_index", C.c_void_p), ("out_xyzl", C.c_void_p),
                ("out_range_origin", C.c_void_p)]


class LrcGathered(C.Structure):
    _fields_ = [("d_all_poses16", C.c_void_p), ("num_poses_all", C.c_uint64), ("d_all_prims", C.c_void_p),
                ("d_all_tile_counts", C.c_void_p), ("poses_per_slab", C.c_uint64), ("slab_stride_bytes", C.c_uint64),
                ("own_slab", C.c_uint64), ("own_ticket", C.c_uint64), ("scan_slot", C.c_uint64), ("d_out_xyzl", C.c_void_p),
                ("d_counts", C.c_void_p)]


class LrcFrames(C.Structure):
    _fields_ = [("counts", C.c_void_p), ("point3", C.c_void_p), ("sem", C.c_void_p), ("ins", C.c_void_p),
                ("incident_deg", C.c_void_p), ("index", C.c_void_p), ("xyzl", C.c_void_p),
                ("range_origin", C.c_void_p), ("range_origin_mean", C.c_void_p), ("range_origin_std", C.c_void_p),
                ("incident_mean", C.c_void_p), ("incident_std", C.c_void_p)]


class LrcMt19937State(C.Structure):
    _fields_ = [("key", C.c_uint32 * 624), ("pos", C.c_int32), ("has_gauss", C.c_int32), ("gauss", C.c_double)]


class LrcGrid(C.Structure):
    _fields_ = [("lines", C.c_uint32), ("width", C.c_uint32), ("az0", C.c_double), ("az_step", C.c_double)]


LRC_STATS_WORDS = 5


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  A PyTorch-ROCm wheel ships its own copy of libamdhip64.so and asks the loader for it under
    that unversioned name, which never matches the SONAME (libamdhip64.so.7) of a copy that is already loaded; this library asks
    for libamdhip64.so.7, which does match torch's copy.  So with torch imported first both share torch's runtime, and with this
    library first the process ends up with two runtimes and torch finds no GPU ("No HIP GPUs are available").  When torch is
    installed but not yet imported, load ITS runtime first (by path, without importing torch): the same arrangement either way.
    LRC_SYSTEM_HIP_RUNTIME=1 keeps the system runtime (a process that will never import torch loses nothing either way)."""
    if "torch" in sys.modules or os.environ.get("LRC_SYSTEM_HIP_RUNTIME") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    runtime = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(runtime):
        try:
            C.CDLL(runtime, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load liblidarcast.so once; raise RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"liblidarcast.so not found at {LIB_PATH}: build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            f"This engine has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, u64, dbl, i32 = C.c_void_p, C.c_uint64, C.c_double, C.c_int
    lib.lrc_version.restype = C.c_char_p
    lib.lrc_version.argtypes = []
    lib.lrc_last_error.restype = C.c_char_p
    lib.lrc_last_error.argtypes = []
    lib.lrc_device_count.restype = i32
    lib.lrc_device_count.argtypes = []
    sig = {
        "lrc_ctx_create": [i32, C.POINTER(vp)],
        "lrc_ctx_destroy": [vp],
        "lrc_ctx_synchronize": [vp],
        "lrc_ctx_set_launch_chaining": [vp, i32],
        "lrc_ctx_get_launch_chaining": [vp, C.POINTER(i32), C.POINTER(i32)],
        "lrc_scene_create": [vp, vp, u64, vp, u64, vp, vp, C.POINTER(vp)],
        "lrc_scene_create_dev": [vp, vp, u64, vp, u64, vp, vp, C.POINTER(vp)],
        "lrc_scene_destroy": [vp],
        "lrc_scene_get_info": [vp, C.POINTER(LrcSceneInfo)],
        "lrc_scene_export_bvh": [vp, vp, vp],
        "lrc_scene_export_array": [vp, i32, vp, u64, C.POINTER(u64)],
        "lrc_scene_get_counters": [vp, C.POINTER(u64), C.POINTER(u64)],
        "lrc_scene_set_options": [vp, C.POINTER(LrcScanOptions)],
        "lrc_scene_get_occupancy": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
        "lrc_cast": [vp, vp, u64, vp, dbl, C.POINTER(LrcHits)],
        "lrc_cast_dev": [vp, vp, u64, vp, dbl, C.POINTER(LrcHits), vp],
        "lrc_cast_segments": [vp, vp, u64, vp, u64, vp, dbl, C.POINTER(LrcHits)],
        "lrc_cast_segments_dev": [vp, vp, u64, vp, u64, vp, dbl, C.POINTER(LrcHits), vp],
        "lrc_scan_poses": [vp, vp, u64, vp, u64, dbl, C.POINTER(LrcHits)],
        "lrc_scan_poses_dev": [vp, vp, u64, vp, u64, dbl, C.POINTER(LrcHits), vp],
        "lrc_scan_poses_compact": [vp, vp, u64, vp, u64, dbl, C.POINTER(LrcFrames), u64, C.POINTER(u64)],
        "lrc_pipe_create": [vp, u64, u64, C.POINTER(vp)],
        "lrc_pipe_destroy": [vp],
        "lrc_pipe_submit": [vp, vp, u64, vp, dbl, C.POINTER(LrcCompactIO), vp, C.POINTER(u64)],
        "lrc_pipe_wait": [vp, vp],
        "lrc_pipe_records": [vp, u64, C.POINTER(LrcHits)],
        "lrc_pipe_trace_ms": [vp, u64, C.POINTER(C.c_float)],
        "lrc_pipe_submit_sharded": [vp, vp, u64, vp, dbl, vp, vp, C.POINTER(LrcGathered), vp, C.POINTER(u64)],
        "lrc_pipe_trace_done": [vp, u64, vp],
        "lrc_pipe_scan_gathered": [vp, vp, C.POINTER(LrcGathered), vp],
        "lrc_pipe_assemble": [vp, vp, C.POINTER(LrcGathered), vp],
        "lrc_host_alloc": [vp, u64, C.POINTER(vp)],
        "lrc_host_free": [vp, vp],
        "lrc_scan_angles_dev": [vp, vp, u64, vp, vp, u64, dbl, C.POINTER(LrcHits), vp],
        "lrc_scan_angles_compact": [vp, vp, u64, vp, vp, u64, dbl, C.POINTER(LrcFrames), u64, C.POINTER(u64)],
        "lrc_debug_scan_stats": [vp, vp, u64, vp, u64, dbl, vp],
        "lrc_scan_rays_compact": [vp, vp, vp, vp, u64, u64, dbl, C.POINTER(LrcFrames), u64, C.POINTER(u64)],
        "lrc_table_create": [vp, vp, u64, C.POINTER(vp)],
        "lrc_table_destroy": [vp],
        "lrc_scan_table_compact": [vp, vp, u64, vp, C.POINTER(LrcGrid), dbl, C.POINTER(LrcFrames), u64, C.POINTER(u64)],
        "lrc_scan_grid_dev": [vp, vp, u64, vp, C.POINTER(LrcGrid), dbl, C.POINTER(LrcHits), vp],
        "lrc_scan_grid_compact": [vp, vp, u64, vp, C.POINTER(LrcGrid), dbl, C.POINTER(LrcFrames), u64, C.POINTER(u64)],
        "lrc_occ_create": [vp, vp, u64, C.POINTER(vp)],
        "lrc_occ_destroy": [vp],
        "lrc_occ_query": [vp, vp, u64, dbl, vp],
        "lrc_min_distances": [vp, vp, u64, vp, u64, vp],
        "lrc_rbf_kernel_sum": [vp, vp, u64, vp, u64, dbl, C.POINTER(dbl)],
        "lrc_nn_create": [vp, vp, u64, dbl, C.POINTER(vp)],
        "lrc_nn_destroy": [vp],
        "lrc_nn_query": [vp, vp, u64, vp, vp],
        "lrc_nn_query_dev": [vp, vp, u64, vp, vp, vp],
        "lrc_cloud_from_ranges_dev": [vp, vp, u64, vp, u64, vp, vp, vp, vp],
        "lrc_cloud_from_prims_dev": [vp, vp, u64, vp, u64, vp, vp, u64, u64, vp, vp, vp],
        "lrc_cloud_from_prims_own_dev": [vp, vp, u64, vp, u64, vp, vp, u64, u64, u64, C.POINTER(LrcCompactIO), vp, vp, vp],
        "lrc_cloud_range_stats_dev": [vp, vp, vp, u64, u64, vp, vp, vp, vp],
        "lrc_compact": [vp, u64, u64, C.POINTER(LrcCompactIO), C.POINTER(u64)],
        "lrc_compact_dev": [vp, u64, u64, C.POINTER(LrcCompactIO), vp],
        "lrc_rng_scan_draws": [C.POINTER(LrcMt19937State), u64, u64, u64, dbl, dbl, vp, vp, i32],
        "lrc_rays_from_trig": [vp, vp, vp, vp, u64, vp, vp],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.restype = i32
        fn.argtypes = argtypes
    _lib = lib
    return lib


class LidarcastError(RuntimeError):
    """A liblidarcast call failed (HIP error, no device, out of memory)."""


def check(rc, what):
    """Map a C status to the Python exception the reference interface promises."""
    if rc == LRC_OK:
        return
    msg = load().lrc_last_error().decode("utf-8", "replace")
    text = f"{what} failed ({rc}): {msg}"
    if rc == LRC_ERR_INVALID_ARG:
        raise ValueError(text)
    if rc == LRC_ERR_OOM:
        raise MemoryError(text)
    raise LidarcastError(text)

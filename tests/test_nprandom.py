"""CPU: the native restatement of numpy's legacy seeded stream (csrc/lrc_nprandom.cpp) against numpy itself -- the
doubles, their order and the generator state afterwards, for the draw pattern of the reference's dual-axis generator
(reference: lidar/indoor_lidar.py:257-296: per pose 2N normals, then N uniforms) and around every edge of the
restatement: block boundaries of the MT19937 state, a cached normal left by an earlier odd draw, odd counts, zero
counts, the global stream and RandomState objects."""
import numpy as np
import pytest


def _numpy_draws(rs, P, nn, nu, loc, scale):
    z = np.empty((P, nn))
    u = np.empty((P, nu))
    for p in range(P):
        z[p] = rs.normal(loc, scale, nn)
        u[p] = rs.random_sample(nu)
    return z, u


def _same_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


@pytest.mark.parametrize("seed,P,nn,nu,pre", [
    (0, 3, 2000, 1000, 0),         # the sensor's pattern, small
    (1, 5, 128, 64, 0),
    (12345, 2, 1, 0, 0),           # odd counts leave a cached normal behind
    (7, 4, 3, 5, 1),               # ... and start from one (pre = normals drawn before)
    (7, 3, 0, 17, 1),              # no normals: the cached value survives the call
    (3, 2, 624, 312, 0),           # exactly block-sized draws
    (3, 6, 311, 1, 623),           # consumption points crossing block ends at every phase
    (99, 1, 0, 0, 0),
    (5, 2, 50001, 25001, 0),       # odd and large
])
@pytest.mark.parametrize("threads", [1, 3])
def test_matches_numpy(seed, P, nn, nu, pre, threads):
    from lidarcast import nprandom
    ref, mine = np.random.RandomState(seed), np.random.RandomState(seed)
    if pre:
        ref.normal(size=pre); mine.normal(size=pre)
        ref.random_sample(3); mine.random_sample(3)
    zr, ur = _numpy_draws(ref, P, nn, nu, 0.25, 1e-3)
    zm, um = nprandom.scan_draws(P, nn, nu, 0.25, 1e-3, rng=mine, threads=threads)
    assert np.array_equal(zr.view(np.uint64), zm.view(np.uint64))
    assert np.array_equal(ur.view(np.uint64), um.view(np.uint64))
    assert _same_state(ref.get_state(), mine.get_state())
    # and the streams stay together afterwards
    assert np.array_equal(ref.normal(size=5), mine.normal(size=5)) and ref.randint(1 << 30) == mine.randint(1 << 30)


@pytest.mark.parametrize("seed,P,nn,nu,pre_words", [
    (0, 7, 20000, 10000, 0),          # even counts, no cached normal, enough work: the streaming path
    (2, 40, 2000, 1000, 1),           # the stream starts one word into a block
    (4, 5, 30000, 14000, 623),        # ... on the last word of a block
    (4, 5, 30000, 14000, 624),        # ... on a block end (numpy regenerates lazily: pos stays 624)
    (6, 4, 600000, 300000, 311),      # poses that span eight 1 MiB chunks: whole chunks skipped by their counts
    (8, 9, 0, 40000, 5),              # uniforms only (sequential path by the shape rule; the comparison still holds)
    (9, 9, 40000, 0, 7),              # normals only
])
@pytest.mark.parametrize("threads", [3, 8])
def test_streaming_path_matches_numpy_and_the_sequential_path(seed, P, nn, nu, pre_words, threads):
    """Even counts and no cached normal: one thread generates words and does nothing else, workers flag every aligned group
    of four words as an accepted polar attempt or not, the caller walks the counts (lrc_nprandom.cpp: scan_streaming).
    The doubles, their order and the generator state afterwards are numpy's; threads < 0 selects the sequential path."""
    from lidarcast import nprandom
    ref, mine, seq = (np.random.RandomState(seed) for _ in range(3))
    for rs in (ref, mine, seq):
        if pre_words:
            rs.randint(0, 1 << 32, size=pre_words, dtype=np.uint32)       # one word each
    assert ref.get_state()[2] == (pre_words % 624 if pre_words else 624) or pre_words == 624
    zr, ur = _numpy_draws(ref, P, nn, nu, -0.5, 2e-3)
    zm, um = nprandom.scan_draws(P, nn, nu, -0.5, 2e-3, rng=mine, threads=threads)
    zs, us = nprandom.scan_draws(P, nn, nu, -0.5, 2e-3, rng=seq, threads=-threads)
    for z, u, rs in ((zm, um, mine), (zs, us, seq)):
        assert np.array_equal(zr.view(np.uint64), z.view(np.uint64))
        assert np.array_equal(ur.view(np.uint64), u.view(np.uint64))
        assert _same_state(ref.get_state(), rs.get_state())
    # a second call continues the stream (now from a mid-block position), then numpy itself does
    zr2, ur2 = _numpy_draws(ref, 2, nn, nu, 0.0, 1.0)
    zm2, um2 = nprandom.scan_draws(2, nn, nu, 0.0, 1.0, rng=mine, threads=threads)
    assert np.array_equal(zr2.view(np.uint64), zm2.view(np.uint64)) and np.array_equal(ur2.view(np.uint64), um2.view(np.uint64))
    assert _same_state(ref.get_state(), mine.get_state())
    assert np.array_equal(ref.normal(size=3), mine.normal(size=3))


def test_global_stream_and_sensor_sized_draw():
    """np.random itself (what the reference uses), one BLK2GO pose: 128 000 normals of sigma 1e-3, 64 000 uniforms."""
    from lidarcast import nprandom
    np.random.seed(0)
    z_ref = np.random.normal(0, 1e-3, size=128000)
    u_ref = np.random.random(64000)
    z2_ref = np.random.normal(0, 1e-3, size=128000)
    after = np.random.get_state()
    np.random.seed(0)
    z, u = nprandom.scan_draws(2, 128000, 64000, 0.0, 1e-3)
    assert np.array_equal(z[0].view(np.uint64), z_ref.view(np.uint64))
    assert np.array_equal(u[0].view(np.uint64), u_ref.view(np.uint64))
    assert np.array_equal(z[1].view(np.uint64), z2_ref.view(np.uint64))
    np.random.random(64000)
    # (the native call drew the second pose's uniforms too: re-seed and compare states pose for pose)
    np.random.seed(0)
    nprandom.scan_draws(1, 128000, 64000, 0.0, 1e-3)
    nprandom.scan_draws(1, 128000, 0, 0.0, 1e-3)
    assert _same_state(np.random.get_state(), after)


def test_only_the_legacy_stream_is_accepted():
    from lidarcast import nprandom
    assert nprandom.supported(None) and nprandom.supported(np.random.RandomState(3))
    assert not nprandom.supported(np.random.default_rng(3))
    with pytest.raises(TypeError):
        nprandom.scan_draws(1, 2, 2, rng=np.random.default_rng(3))


@pytest.mark.parametrize("seed", [0, 12345])
def test_trajectory_rays_match_per_pose_generator(seed):
    """The batched host generator of the dual-axis sensor (native draws for runs of poses + thread pool) against
    ``get_rays()`` called pose after pose on the same seed: identical rays, identical masks, identical generator state
    afterwards -- for the global stream and for a private RandomState."""
    import dataclasses
    from lidar import DualAxisLidar, DualAxisLidarIntrinsics
    from raycast_engine.raycast_engine_hip import dual_axis_rays_batch
    from helpers import pose
    k = dataclasses.replace(DualAxisLidarIntrinsics.create_blk2go_dual_axis(), point_rate=64000.0)   # 6 400 rays per pose
    poses = [pose(1.0 + 0.1 * i, 2.0, 1.0, yaw=0.3 * i) for i in range(37)]
    n = k.num_vertical_lines * (int(k.point_rate * k.scan_duration) // k.num_vertical_lines)
    for private in (False, True):
        rs = np.random.RandomState(seed) if private else None
        np.random.seed(seed)
        ref = [DualAxisLidar(k, m, rng=rs).get_rays() for m in poses]
        state_ref = rs.get_state() if private else np.random.get_state()
        rs = np.random.RandomState(seed) if private else None
        np.random.seed(seed)
        lidars = [DualAxisLidar(k, m, rng=rs) for m in poses]
        rays = np.empty((len(poses), n, 6), dtype=np.float32)
        keep = np.ones((len(poses), n), dtype=np.uint8)
        dual_axis_rays_batch(lidars, rays, keep)
        state = rs.get_state() if private else np.random.get_state()
        assert _same_state(state_ref, state)
        for i in range(len(poses)):
            got = rays[i][keep[i].astype(bool)]
            assert got.shape == ref[i].shape and np.array_equal(got.view(np.uint32), ref[i].view(np.uint32)), i


def test_native_ray_composition_equals_the_numpy_formula():
    """lrc_rays_from_trig (products, rotation, narrowing of a dual-axis pose in one native pass) against
    IndoorLidar.rays_from_angles, the numpy form of the reference's arithmetic (lidar/indoor_lidar.py:274-291): the same
    float32 bits for a pose rotated about all three axes, angles over the whole range, non-finite values passed through."""
    import ctypes as C
    from lidarcast import _capi
    from lidar import DualAxisLidarIntrinsics, create_lidar
    lib = _capi.load()
    rng = np.random.default_rng(11)
    n = 10007
    phi = rng.uniform(-7.0, 7.0, n)
    theta = rng.uniform(-1.6, 1.6, n)
    phi[:3] = [0.0, np.pi, -0.0]
    theta[3:6] = [np.pi / 2, -np.pi / 2, 0.0]
    a, b, c = 0.7, -0.31, 1.9
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
    pose = np.eye(4)
    pose[:3, :3] = Rz @ Ry @ Rx
    pose[:3, 3] = [1.2345678901, -7.000000123, 0.333333333333]
    lidar = create_lidar(DualAxisLidarIntrinsics.create_blk2go_dual_axis(), pose)
    ref = lidar.rays_from_angles(phi, theta)
    out = np.full((n, 6), np.nan, dtype=np.float32)
    ct, st, cp, sp = np.cos(theta), np.sin(theta), np.cos(phi), np.sin(phi)
    M = np.ascontiguousarray(pose, dtype=np.float64)
    _capi.check(lib.lrc_rays_from_trig(ct.ctypes.data, st.ctypes.data, cp.ctypes.data, sp.ctypes.data, n, M.ctypes.data,
                                       out.ctypes.data), "lrc_rays_from_trig")
    assert np.array_equal(ref.view(np.uint32), out.view(np.uint32))
    assert lib.lrc_rays_from_trig(None, None, None, None, 0, None, None) == 0
    assert lib.lrc_rays_from_trig(None, st.ctypes.data, cp.ctypes.data, sp.ctypes.data, n, M.ctypes.data, out.ctypes.data) != 0


@pytest.mark.parametrize("budget", [0, 1, 2, 4])
@pytest.mark.parametrize("threads", [8, -4])
def test_thread_start_failures_leave_fewer_threads_not_a_dead_process(budget, threads):
    """a std::thread that cannot start (EAGAIN at a thread limit) must not unwind past joinable threads
    (std::terminate).  The hook makes the first `budget` starts succeed and every later one fail: no generator thread
    -> the sequential path; some workers -> fewer workers; the draws and the state stay numpy's."""
    import ctypes
    import lidarcast
    from lidarcast import nprandom
    lib = lidarcast.load()
    hook = lib.lrc_internal_set_thread_budget
    hook.argtypes, hook.restype = [ctypes.c_long], None
    P, nn, nu = 6, 20000, 10000
    ref, mine = np.random.RandomState(21), np.random.RandomState(21)
    zr, ur = _numpy_draws(ref, P, nn, nu, 0.0, 1e-3)
    hook(budget)
    try:
        zm, um = nprandom.scan_draws(P, nn, nu, 0.0, 1e-3, rng=mine, threads=threads)
    finally:
        hook(-1)
    assert np.array_equal(zr.view(np.uint64), zm.view(np.uint64))
    assert np.array_equal(ur.view(np.uint64), um.view(np.uint64))
    assert _same_state(ref.get_state(), mine.get_state())

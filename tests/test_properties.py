"""Property tests (hypothesis): adversarial small scenes -- snapped coordinates (axis-aligned, coincident and
degenerate triangles), rays with zero direction components, origins on vertices/planes -- where the closest hit
must be the same for brute force (the definition), the oracle's own BVH and, on the GPU, the HIP traversal."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from helpers import assert_bit_equal
from oracle.c_oracle import OracleMesh

GRID = [-1.0, -0.5, 0.0, 0.25, 0.5, 1.0, 2.0]


@st.composite
def scenes(draw):
    n_tris = draw(st.integers(1, 40))
    snap = draw(st.booleans())
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    if snap:
        v = rng.choice(GRID, size=(n_tris * 3, 3))
    else:
        v = rng.uniform(-1, 2, size=(n_tris * 3, 3))
        v[rng.random(len(v)) < 0.3] = rng.choice(GRID, size=3)          # some shared / snapped vertices
    f = np.arange(n_tris * 3).reshape(-1, 3)
    if draw(st.booleans()):                                              # re-use vertices across triangles
        f = rng.integers(0, len(v), size=(n_tris, 3))
    n_rays = 200
    o = rng.choice(GRID, size=(n_rays, 3)) if draw(st.booleans()) else rng.uniform(-1.5, 2.5, size=(n_rays, 3))
    d = rng.normal(size=(n_rays, 3))
    axis = rng.random(n_rays) < 0.4
    d[axis] = rng.choice([-1.0, 0.0, 1.0], size=(int(axis.sum()), 3))     # axis-parallel, diagonal and zero directions
    tgt = rng.random(n_rays) < 0.3                                        # aimed exactly at a vertex
    d[tgt] = v[rng.integers(0, len(v), int(tgt.sum()))] - o[tgt]
    return v, f.astype(np.int32), np.concatenate([o, d], 1).astype(np.float32)


@settings(max_examples=int(os.environ.get("LRC_HYPOTHESIS_EXAMPLES", 60)), deadline=None, derandomize="LRC_HYPOTHESIS_EXAMPLES" not in os.environ,
          suppress_health_check=list(HealthCheck))
@given(scenes())
def test_oracle_bvh_equals_definition(scene):
    v, f, rays = scene
    om = OracleMesh(v, f)
    tb, pb = om.brute(rays)
    t, p = om.cast(rays)
    assert_bit_equal(t, tb)
    assert_bit_equal(p, pb)
    hit = np.isfinite(tb)
    assert (tb[hit] > 0).all() and (pb[~hit] == 0xFFFFFFFF).all() and (pb[hit] < len(f)).all()


@pytest.mark.gpu
@settings(max_examples=int(os.environ.get("LRC_HYPOTHESIS_EXAMPLES", 40)), deadline=None, derandomize="LRC_HYPOTHESIS_EXAMPLES" not in os.environ,
          suppress_health_check=list(HealthCheck))
@given(scenes())
def test_hip_equals_definition(scene):
    import lidarcast
    global _CTX
    try:
        _CTX
    except NameError:
        _CTX = lidarcast.Context(0)
    v, f, rays = scene
    om = OracleMesh(v, f)
    tb, pb = om.brute(rays)
    sc = lidarcast.Scene(_CTX, v, f)
    out = sc.cast(rays, want=("t", "prim", "normal3"))
    sc.close()
    assert_bit_equal(out["t"], tb)
    assert_bit_equal(out["prim"], pb)
    assert_bit_equal(out["normal3"], om.normals(pb))


@pytest.mark.gpu
@settings(max_examples=int(os.environ.get("LRC_HYPOTHESIS_EXAMPLES", 25)), deadline=None, derandomize="LRC_HYPOTHESIS_EXAMPLES" not in os.environ,
          suppress_health_check=list(HealthCheck))
@given(scenes(), st.integers(1, 4), st.booleans())
def test_cloud_rebuilt_from_ids_on_adversarial_scenes(scene, num_poses, aligned):
    """Pose-batched scan of the same adversarial scenes (the rays' directions become the sensor's direction table,
    poses are yawed and snapped): the scan equals the definition on the rays the host generator gives, and the cloud
    rebuilt from nothing but the 4-byte triangle ids equals the compaction of the scan's own records, bit for bit --
    ties, grazing hits and degenerate triangles included."""
    import torch
    import lidarcast
    from lidarcast._capi import LrcCompactIO
    from helpers import pose
    global _CTX
    try:
        _CTX
    except NameError:
        _CTX = lidarcast.Context(0)
    v, f, rays = scene
    dirs = np.ascontiguousarray(rays[:192 if aligned else 200, 3:].astype(np.float64))   # 192 = 3 aligned tiles
    rng = np.random.default_rng(len(v) * 7 + num_poses)
    poses = np.stack([pose(*rng.choice(GRID, 3), yaw=float(rng.choice([0.0, 0.5, np.pi / 2, -2.0])))
                      for _ in range(num_poses)])
    P, N = len(poses), len(dirs)
    sc = lidarcast.Scene(_CTX, v, f)
    dev = torch.device("cuda", 0)
    st_ = torch.cuda.current_stream().cuda_stream
    want = ("t", "prim", "point3", "sem", "ins") + (("tile_count",) if aligned else ())
    hits = lidarcast.DeviceHits(P * N, dev, want=want)
    d_poses, d_dirs = torch.from_numpy(poses.reshape(P, 16)).to(dev), torch.from_numpy(dirs).to(dev)
    sc.scan_poses_dev(d_poses, d_dirs, hits, 3.0, st_)
    rows = torch.full((P * N, 4), 7.0, dtype=torch.float32, device=dev)
    counts = torch.zeros(P, dtype=torch.int64, device=dev)
    io = LrcCompactIO()
    io.t, io.point3, io.sem, io.ins = (hits[a].data_ptr() for a in ("t", "point3", "sem", "ins"))
    io.counts, io.out_xyzl = counts.data_ptr(), rows.data_ptr()
    _CTX.compact_dev(P, N, io, st_)
    rows2, counts2 = torch.full_like(rows, 7.0), torch.zeros_like(counts)
    sc.cloud_from_prims_dev(d_poses, d_dirs, hits["prim"], rows2, counts2,
                            tile_count_t=hits["tile_count"] if aligned else None, stream=st_)
    torch.cuda.synchronize()
    assert torch.equal(counts2, counts)
    assert torch.equal(rows2.view(torch.int32), rows.view(torch.int32))
    # and the scan itself against the definition, on the rays the host generator produces for these poses
    om = OracleMesh(v, f)
    t_gpu, prim_gpu = hits["t"].cpu().numpy().reshape(P, N), hits["prim"].cpu().numpy().view(np.uint32).reshape(P, N)
    for p in range(P):
        o = np.repeat(poses[p][:3, 3][None, :], N, 0).astype(np.float32)
        d = np.dot(dirs, poses[p][:3, :3].T).astype(np.float32)
        tb, pb = om.brute(np.concatenate([o, d], 1))
        with np.errstate(invalid="ignore", divide="ignore"):      # hypothesis also draws zero-length directions
            pts = o + (d / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])[:, None]) * np.where(np.isfinite(tb), tb, 0)[:, None]
            keep = np.isfinite(tb) & (np.linalg.norm(pts.astype(np.float64) - poses[p][:3, 3], axis=1) < 3.0)
        assert_bit_equal(t_gpu[p], np.where(keep, tb, np.inf).astype(np.float32))
        assert_bit_equal(prim_gpu[p], np.where(keep, pb, 0xFFFFFFFF).astype(np.uint32))
    sc.close()


@pytest.mark.gpu
@settings(max_examples=int(os.environ.get("LRC_HYPOTHESIS_EXAMPLES", 25)), deadline=None, derandomize="LRC_HYPOTHESIS_EXAMPLES" not in os.environ,
          suppress_health_check=list(HealthCheck))
@given(scenes(), st.integers(1, 5), st.sampled_from([0.8, 2.0, 50.0]))
def test_frames_and_statistics_on_adversarial_scenes(scene, num_poses, max_range):
    """lrc_scan_poses_compact on the adversarial scenes: the kept rows of every pose equal the fixed-stride records of
    lrc_scan_poses masked on the host (ties, grazing hits, zero directions, range filter cutting through), and the
    per-pose statistics equal np.mean / np.std of those rows bit for bit, for whatever frame sizes come out."""
    import lidarcast
    from helpers import pose
    global _CTX
    try:
        _CTX
    except NameError:
        _CTX = lidarcast.Context(0)
    v, f, rays = scene
    dirs = np.ascontiguousarray(rays[:, 3:].astype(np.float64))
    rng = np.random.default_rng(len(v) * 13 + num_poses)
    poses = np.stack([pose(*rng.choice(GRID, 3), yaw=float(rng.choice([0.0, 0.5, np.pi / 2, -2.0])))
                      for _ in range(num_poses)])
    sc = lidarcast.Scene(_CTX, v, f)
    rec = sc.scan_poses(poses, dirs, max_range, want=("t", "point3", "incident_deg", "sem", "ins"))
    fr = sc.scan_poses_compact(poses, dirs, max_range, want=("point3", "incident_deg", "index", "range_origin",
                                                             "range_origin_stats", "incident_stats"))
    sc.close()
    P, N = len(poses), len(dirs)
    keep = np.isfinite(rec["t"]).reshape(P, N)
    assert np.array_equal(fr["counts"], keep.sum(1)) and fr["total"] == keep.sum()
    assert_bit_equal(fr["point3"], rec["point3"].reshape(P, N, 3)[keep])
    assert_bit_equal(fr["incident_deg"], rec["incident_deg"].reshape(P, N)[keep])
    assert np.array_equal(fr["index"], np.concatenate([np.flatnonzero(m) for m in keep]).astype(np.uint32))
    ends = np.cumsum(fr["counts"])
    for i in range(P):
        r, a = fr["range_origin"][ends[i] - fr["counts"][i]:ends[i]], fr["incident_deg"][ends[i] - fr["counts"][i]:ends[i]]
        if len(r) == 0:
            assert fr["range_origin_mean"][i] == 0 and fr["incident_std"][i] == 0
            continue
        assert_bit_equal(r, np.linalg.norm(fr["point3"][ends[i] - fr["counts"][i]:ends[i]], axis=1))
        assert_bit_equal(fr["range_origin_mean"][i:i + 1], np.array([np.mean(r)]))
        assert_bit_equal(fr["range_origin_std"][i:i + 1], np.array([np.std(r)]))
        with np.errstate(invalid="ignore"):
            am, asd = np.mean(a), np.std(a)
        if np.isfinite(am) and np.isfinite(asd):
            assert_bit_equal(fr["incident_mean"][i:i + 1], np.array([am]))
            assert_bit_equal(fr["incident_std"][i:i + 1], np.array([asd]))

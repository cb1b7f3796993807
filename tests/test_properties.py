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

"""Row N2: the trajectory planner against outputs of the reference's own AutoTrajectoryGenerator
(tests/golden/make_planner_golden.py), same np.random seed: identical free-space grid, graph, candidates and
waypoints, bit for bit.  GPU test: the robot-cube occupancy runs as a HIP kernel."""
import os

import numpy as np
import pytest

from conftest import REPO
from helpers import assert_bit_equal


@pytest.fixture(scope="module")
def pg():
    return np.load(os.path.join(REPO, "tests", "golden", "planner_golden.npz"))


class _Mesh:
    def __init__(self, v):
        self.vertices = v
        self.triangles = np.zeros((0, 3), np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_planner_reproduces_reference(pg, tag):
    from trajectory import AutoTrajectoryGenerator
    v = pg[f"{tag}_vertices"]
    keys = ("x_min", "x_max", "y_min", "y_max", "z_min", "z_max")
    bounds = dict(zip(keys, map(float, pg[f"{tag}_bounds"])))
    seed, n, radius = pg[f"{tag}_params"]
    gen = AutoTrajectoryGenerator(robot_radius=float(radius))
    np.random.seed(int(seed))
    wps, info = gen.generate_optimal_trajectory(_Mesh(v), bounds, num_waypoints=int(n))
    ra = gen.room_analysis
    assert_bit_equal(np.array(ra.free_space_points), pg[f"{tag}_free"], "free-space grid")
    assert_bit_equal(np.array(ra.obstacle_points).reshape(-1, 3), pg[f"{tag}_blocked"], "blocked grid points")
    assert np.array_equal([len(ra.connectivity_graph[i]) for i in range(len(ra.free_space_points))], pg[f"{tag}_degree"])
    got = np.array([[w.x, w.y, w.z, w.yaw] for w in wps], dtype=np.float64)
    assert_bit_equal(got, pg[f"{tag}_waypoints"], "waypoints")
    b = info["best_trajectory"]
    s = pg[f"{tag}_summary"]
    assert info["total_candidates"] == s[0] and b["length"] == s[1] and b["collision_count"] == s[2]
    assert b["smoothness_score"] == s[3] and info["statistics"]["length_mean"] == s[4]
    assert info["statistics"]["collision_mean"] == s[5] and gen.min_trajectory_length == s[6]
    assert float(np.random.random()) == float(pg[f"{tag}_next_uniform"])          # same number of draws consumed
    assert len(wps) == max(int(n) * 2, 40) and all(w.yaw == 0 and w.z == 1.0 for w in wps)


@pytest.mark.gpu
def test_occupancy_kernel_matches_numpy():
    import lidarcast
    ctx = lidarcast.Context(0)
    rng = np.random.default_rng(0)
    v = rng.uniform(0, 5, (50_000, 3))
    q = rng.uniform(-0.5, 5.5, (700, 3))
    q[:5] = v[:5]                                        # exactly on a vertex
    q[5] = v[5] + [0.15, 0, 0]                           # exactly on the cube face: inclusive
    occ = lidarcast.OccupancyIndex(ctx, v)
    for half in (0.15, 0.05, 1e-9):
        want = np.array([np.any(np.all((v >= p - half) & (v <= p + half), axis=1)) for p in q])
        assert np.array_equal(occ.occupied(q, half), want)
    assert not lidarcast.OccupancyIndex(ctx, np.zeros((0, 3))).occupied(q, 0.2).any()
    assert occ.occupied(np.zeros((0, 3)), 0.2).shape == (0,)


def test_waypoint_helpers_and_missing_reference_names():
    """CPU: the names the reference's simulator imports from ``trajectory`` exist."""
    from trajectory import PathType, SmartTrajectoryGenerator, TrajectoryQuality, Waypoint
    g = SmartTrajectoryGenerator({"x_min": 0.0, "x_max": 5.0, "y_min": 0.0, "y_max": 4.0, "z_min": 0.0, "z_max": 3.0},
                                 robot_height=1.0)
    wps, q = g.generate_trajectory((1, 2, 1), (4, 2, 1), PathType.STRAIGHT, 7)
    assert len(wps) == 7 and isinstance(q, TrajectoryQuality) and abs(q.path_length - 3.0) < 1e-12
    assert isinstance(wps[0], Waypoint) and wps[-1].x == 4 and q.to_dict()["collision_count"] == 0

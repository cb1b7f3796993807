"""TrajectoryGeneratorBase measures against values from the reference's own class
(tests/golden/make_trajectory_base_golden.py), exact."""
import json
import os

import numpy as np
import pytest

from trajectory import PathType, SmartTrajectoryGenerator, TrajectoryGeneratorBase, Waypoint

HERE = os.path.dirname(os.path.abspath(__file__))


class Gen(TrajectoryGeneratorBase):
    def generate_trajectory(self, **kwargs):
        return [], None


def test_base_class_measures_match_the_reference():
    with open(os.path.join(HERE, "golden", "trajectory_base_golden.json")) as f:
        G = json.load(f)
    g = Gen(G["bounds"], robot_height=G["robot_height"])
    assert g.robot_radius == G["robot_radius"]
    for name, c in G["cases"].items():
        wps = [Waypoint(*r) for r in c["waypoints"]]
        assert float(g.calculate_path_length(wps)) == c["path_length"], name
        assert g.count_turns(wps) == c["turns_default"] and g.count_turns(wps, 1.0) == c["turns_1p0"], name
        assert float(g.calculate_smoothness(wps)) == c["smoothness"], name
        assert float(g._calculate_coverage_ratio(wps)) == c["coverage"], name
        assert [bool(g.is_point_in_room(w)) for w in wps] == c["in_room"], name
        got = [[float(k.x), float(k.y), float(k.z), float(k.yaw), float(k.timestamp), float(k.velocity),
                float(k.angular_velocity)] for k in (g.clip_to_room_bounds(w) for w in wps)]
        assert got == c["clipped"], name
        q = g.evaluate_trajectory_quality(wps, collision_count=len(wps) % 3)
        assert {k: float(v) for k, v in q.to_dict().items()} == c["quality"], name
        assert [m.tolist() for m in g.waypoints_to_poses(wps)] == c["poses"], name


def test_abstract_and_stand_in_generator():
    with pytest.raises(TypeError):
        TrajectoryGeneratorBase({"x_min": 0, "x_max": 1, "y_min": 0, "y_max": 1, "z_min": 0, "z_max": 1})
    b = {"x_min": 0.0, "x_max": 5.0, "y_min": 0.0, "y_max": 4.0, "z_min": 0.0, "z_max": 3.0}
    gen = SmartTrajectoryGenerator(b, robot_height=1.0)
    wps, q = gen.generate_trajectory((1.0, 2.0, 1.0), (4.0, 2.0, 1.0), PathType.STRAIGHT, 7)
    assert len(wps) == 7 and np.isclose(q.path_length, 3.0) and q.turn_count == 0 and q.smoothness == 1.0
    assert q.coverage_ratio == 0.0 and q.efficiency == 0.0      # a straight line spans no area

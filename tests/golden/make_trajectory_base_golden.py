#!/usr/bin/env python3
"""Golden values of TrajectoryGeneratorBase's measures, from the REFERENCE's own trajectory/trajectory_generator.py
(numpy only, loaded standalone; build container only).

    python tests/golden/make_trajectory_base_golden.py    # writes tests/golden/trajectory_base_golden.json
"""
import importlib.util
import json
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_trajectory_generator",
                                              os.path.join(REF, "trajectory", "trajectory_generator.py"))
tg = importlib.util.module_from_spec(spec)
sys.modules["ref_trajectory_generator"] = tg
spec.loader.exec_module(tg)


class Gen(tg.TrajectoryGeneratorBase):
    def generate_trajectory(self, **kwargs):
        return [], None


def main():
    rng = np.random.RandomState(9)
    bounds = {"x_min": -1.0, "x_max": 5.0, "y_min": 0.0, "y_max": 4.0, "z_min": 0.0, "z_max": 3.0}
    g = Gen(bounds, robot_height=1.2)
    cases = {}
    for name, n in (("empty", 0), ("one", 1), ("two", 2), ("three", 3), ("many", 25)):
        rows = [[float(v) for v in (rng.uniform(-2, 6), rng.uniform(-1, 5), rng.uniform(-0.5, 3.5),
                                    rng.uniform(-3.5, 3.5), float(i), rng.uniform(0, 1), rng.uniform(-1, 1))]
                for i in range(n)]
        wps = [tg.Waypoint(*r) for r in rows]
        q = g.evaluate_trajectory_quality(wps, collision_count=n % 3)
        cases[name] = {
            "waypoints": rows,
            "path_length": float(g.calculate_path_length(wps)),
            "turns_default": int(g.count_turns(wps)), "turns_1p0": int(g.count_turns(wps, 1.0)),
            "smoothness": float(g.calculate_smoothness(wps)),
            "coverage": float(g._calculate_coverage_ratio(wps)),
            "in_room": [bool(g.is_point_in_room(w)) for w in wps],
            "clipped": [[float(c.x), float(c.y), float(c.z), float(c.yaw), float(c.timestamp), float(c.velocity),
                         float(c.angular_velocity)] for c in (g.clip_to_room_bounds(w) for w in wps)],
            "quality": {k: float(v) for k, v in q.to_dict().items()},
            "poses": [m.tolist() for m in g.waypoints_to_poses(wps)],
        }
    out = {"bounds": bounds, "robot_height": g.robot_height, "robot_radius": g.robot_radius, "cases": cases}
    with open(os.path.join(OUT, "trajectory_base_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", list(cases))


if __name__ == "__main__":
    main()

"""Waypoint -> sensor pose, the trajectory quality record and the generator base class
(reference: trajectory/trajectory_generator.py; results checked against the reference's own class,
tests/golden/make_trajectory_base_golden.py).  The planner itself is auto_trajectory_generator.py.
"""
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np


@dataclass
class Waypoint:
    """Position + yaw (rad) of the sensor; pose = translation and a rotation about +Z."""
    x: float
    y: float
    z: float
    yaw: float
    timestamp: float = 0.0
    velocity: Optional[float] = None
    angular_velocity: Optional[float] = None

    def to_array(self) -> np.ndarray:
        return np.array([self.x, self.y, self.z, self.yaw])

    def to_pose_matrix(self) -> np.ndarray:
        m = np.eye(4)
        m[:3, 3] = (self.x, self.y, self.z)
        c, s = np.cos(self.yaw), np.sin(self.yaw)
        m[0, 0], m[0, 1] = c, -s
        m[1, 0], m[1, 1] = s, c
        return m

    def distance_to(self, other: "Waypoint") -> float:
        return np.sqrt((self.x - other.x) ** 2 + (self.y - other.y) ** 2 + (self.z - other.z) ** 2)

    def angle_to(self, other: "Waypoint") -> float:
        return np.arctan2(other.y - self.y, other.x - self.x)

    def __repr__(self) -> str:
        return f"Waypoint(x={self.x:.2f}, y={self.y:.2f}, z={self.z:.2f}, yaw={self.yaw:.2f})"


@dataclass
class TrajectoryQuality:
    """Quality record of a planned trajectory (reference: trajectory/trajectory_generator.py:64-81)."""
    coverage_ratio: float
    path_length: float
    turn_count: int
    efficiency: float
    collision_count: int
    smoothness: float

    def to_dict(self):
        return {"coverage_ratio": self.coverage_ratio, "path_length": self.path_length,
                "turn_count": self.turn_count, "efficiency": self.efficiency,
                "collision_count": self.collision_count, "smoothness": self.smoothness}


def _yaw_steps(waypoints: List[Waypoint]) -> List[float]:
    """|yaw[i+1] - yaw[i]| for the interior waypoints i = 1 .. n-2, folded into [0, pi]."""
    out = []
    for i in range(1, len(waypoints) - 1):
        step = abs(waypoints[i + 1].yaw - waypoints[i].yaw)
        out.append(2 * np.pi - step if step > np.pi else step)
    return out


class TrajectoryGeneratorBase(ABC):
    """Common measures of a waypoint list (reference: trajectory/trajectory_generator.py:84-241)."""

    def __init__(self, room_bounds: Dict[str, float], robot_height: float = 1.0):
        self.room_bounds = room_bounds
        self.robot_height = robot_height
        self.robot_radius = 0.3

    @abstractmethod
    def generate_trajectory(self, **kwargs) -> Tuple[List[Waypoint], TrajectoryQuality]:
        """(waypoints, quality)"""

    def waypoints_to_poses(self, waypoints: List[Waypoint]) -> List[np.ndarray]:
        return [w.to_pose_matrix() for w in waypoints]

    def calculate_path_length(self, waypoints: List[Waypoint]) -> float:
        total = 0.0
        for prev, cur in zip(waypoints, waypoints[1:]):
            total += cur.distance_to(prev)
        return total

    def count_turns(self, waypoints: List[Waypoint], angle_threshold: float = 0.1) -> int:
        """Interior waypoints after which the yaw changes by more than the threshold."""
        return sum(1 for step in _yaw_steps(waypoints) if step > angle_threshold)

    def calculate_smoothness(self, waypoints: List[Waypoint]) -> float:
        """1 / (1 + std of the yaw steps); 1.0 for fewer than three waypoints."""
        steps = _yaw_steps(waypoints)
        return 1.0 / (1.0 + np.std(steps)) if steps else 1.0

    def is_point_in_room(self, waypoint: Waypoint) -> bool:
        b = self.room_bounds
        return (b["x_min"] <= waypoint.x <= b["x_max"] and b["y_min"] <= waypoint.y <= b["y_max"]
                and b["z_min"] <= waypoint.z <= b["z_max"])

    def clip_to_room_bounds(self, waypoint: Waypoint) -> Waypoint:
        b = self.room_bounds
        return Waypoint(x=np.clip(waypoint.x, b["x_min"], b["x_max"]), y=np.clip(waypoint.y, b["y_min"], b["y_max"]),
                        z=np.clip(waypoint.z, b["z_min"], b["z_max"]), yaw=waypoint.yaw,
                        timestamp=waypoint.timestamp, velocity=waypoint.velocity,
                        angular_velocity=waypoint.angular_velocity)

    def _calculate_coverage_ratio(self, waypoints: List[Waypoint]) -> float:
        """Area of the waypoints' x/y bounding rectangle over the room's floor area, capped at 1."""
        if not waypoints:
            return 0.0
        xy = np.array([[w.x, w.y] for w in waypoints])
        span = xy.max(axis=0) - xy.min(axis=0)
        b = self.room_bounds
        floor = (b["x_max"] - b["x_min"]) * (b["y_max"] - b["y_min"])
        return min(span[0] * span[1] / floor, 1.0)

    def evaluate_trajectory_quality(self, waypoints: List[Waypoint], collision_count: int = 0) -> TrajectoryQuality:
        length = self.calculate_path_length(waypoints)
        coverage = self._calculate_coverage_ratio(waypoints)
        return TrajectoryQuality(coverage_ratio=coverage, path_length=length, turn_count=self.count_turns(waypoints),
                                 efficiency=coverage / length if length > 0 else 0,
                                 collision_count=collision_count, smoothness=self.calculate_smoothness(waypoints))


def poses_from_waypoints(waypoints: List[Waypoint]) -> np.ndarray:
    """(P,4,4) float64 stack of ``to_pose_matrix()``."""
    n = len(waypoints)
    if n == 0:
        return np.zeros((0, 4, 4))
    # one array assembly instead of n small ones; the trigonometry is still evaluated per DISTINCT yaw with the scalar
    # np.cos / np.sin calls of to_pose_matrix (auto trajectories have a single yaw, trajectory/auto_trajectory_generator.py)
    out = np.zeros((n, 4, 4))
    out[:, 0, 0] = out[:, 1, 1] = out[:, 2, 2] = out[:, 3, 3] = 1.0
    out[:, 0, 3] = [w.x for w in waypoints]
    out[:, 1, 3] = [w.y for w in waypoints]
    out[:, 2, 3] = [w.z for w in waypoints]
    trig = {}
    cs = np.empty((n, 2))
    for i, w in enumerate(waypoints):
        t = trig.get(w.yaw)
        if t is None:
            t = trig[w.yaw] = (np.cos(w.yaw), np.sin(w.yaw))
        cs[i] = t
    out[:, 0, 0], out[:, 0, 1] = cs[:, 0], -cs[:, 1]
    out[:, 1, 0], out[:, 1, 1] = cs[:, 1], cs[:, 0]
    return out


def line_trajectory(start, end, num_waypoints: int, yaw: float = 0.0) -> List[Waypoint]:
    """Evenly spaced, pure-translation waypoints: the shape auto trajectories have in the reference
    (trajectory/auto_trajectory_generator.py:61-62,83,122 -- every waypoint yaw=0, fixed height)."""
    s, e = np.asarray(start, dtype=np.float64), np.asarray(end, dtype=np.float64)
    ts = np.linspace(0.0, 1.0, int(num_waypoints)) if num_waypoints > 1 else np.array([0.0])
    return [Waypoint(*(s + t * (e - s)), yaw=yaw, timestamp=float(i)) for i, t in enumerate(ts)]

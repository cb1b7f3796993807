"""Waypoint -> sensor pose (reference: trajectory/trajectory_generator.py:12-61).

Only the pose source of the scan path is provided here; trajectory planning is out of scope
(SURVEY.md section 8(f), row N2).
"""
from dataclasses import dataclass
from typing import List, Optional

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


def poses_from_waypoints(waypoints: List[Waypoint]) -> np.ndarray:
    """(P,4,4) float64 stack of ``to_pose_matrix()``."""
    if len(waypoints) == 0:
        return np.zeros((0, 4, 4))
    return np.stack([w.to_pose_matrix() for w in waypoints])


def line_trajectory(start, end, num_waypoints: int, yaw: float = 0.0) -> List[Waypoint]:
    """Evenly spaced, pure-translation waypoints: the shape auto trajectories have in the reference
    (trajectory/auto_trajectory_generator.py:61-62,83,122 -- every waypoint yaw=0, fixed height)."""
    s, e = np.asarray(start, dtype=np.float64), np.asarray(end, dtype=np.float64)
    ts = np.linspace(0.0, 1.0, int(num_waypoints)) if num_waypoints > 1 else np.array([0.0])
    return [Waypoint(*(s + t * (e - s)), yaw=yaw, timestamp=float(i)) for i, t in enumerate(ts)]

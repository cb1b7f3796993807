"""Pose source of the scan path (import surface of the reference's ``trajectory`` package that the scan needs)."""
from enum import Enum

from .trajectory_generator import (TrajectoryGeneratorBase, Waypoint, TrajectoryQuality, poses_from_waypoints,
                                   line_trajectory)
from .auto_trajectory_generator import AutoTrajectoryGenerator, RoomAnalysis, TrajectoryCandidate


class PathType(Enum):
    """The reference's simulator imports this name from ``trajectory`` although the reference never defines it
    (SURVEY.md F8); only the straight path it defaults to (s3dis_simulator.py:182) is provided."""
    STRAIGHT = "straight"


class SmartTrajectoryGenerator(TrajectoryGeneratorBase):
    """Also imported but undefined in the reference.  Minimal stand-in with the call shape the
    simulator uses (s3dis_simulator.py:124-127, :201-206): straight line between two poses at fixed yaw, scored by
    the base class's measures."""

    def __init__(self, room_bounds, robot_height: float = 1.0):
        super().__init__(room_bounds, robot_height)
        self.collision_detector = None

    def generate_trajectory(self, start_point, end_point, path_type=PathType.STRAIGHT, num_waypoints: int = 20):
        if path_type is not PathType.STRAIGHT:
            raise ValueError("only PathType.STRAIGHT is available")
        wps = line_trajectory(start_point, end_point, num_waypoints)
        return wps, self.evaluate_trajectory_quality(wps)


__all__ = ["TrajectoryGeneratorBase", "Waypoint", "TrajectoryQuality", "poses_from_waypoints", "line_trajectory", "AutoTrajectoryGenerator",
           "RoomAnalysis", "TrajectoryCandidate", "PathType", "SmartTrajectoryGenerator"]

"""Pose source of the scan path (subset of the reference's ``trajectory`` package)."""
from .trajectory_generator import Waypoint, poses_from_waypoints, line_trajectory

__all__ = ["Waypoint", "poses_from_waypoints", "line_trajectory"]

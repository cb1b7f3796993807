"""Sensor models of the scan path (import surface of the reference's ``lidar`` package)."""
from .lidar_intrinsics import LidarIntrinsics, Indoor8LineLidarIntrinsics, DualAxisLidarIntrinsics
from .indoor_lidar import IndoorLidar, DualAxisLidar, create_lidar, get_lidar_type

__all__ = ["LidarIntrinsics", "Indoor8LineLidarIntrinsics", "DualAxisLidarIntrinsics",
           "IndoorLidar", "DualAxisLidar", "create_lidar", "get_lidar_type"]

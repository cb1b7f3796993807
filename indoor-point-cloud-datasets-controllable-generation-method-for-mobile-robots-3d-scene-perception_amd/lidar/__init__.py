"""Sensor models of the scan path: parameter records, vectorised ray generators, the create_lidar factory.
Names match what callers of the reference import from its ``lidar`` package."""
from . import indoor_lidar as _gen
from . import lidar_intrinsics as _par

LidarIntrinsics = _par.LidarIntrinsics
Indoor8LineLidarIntrinsics = _par.Indoor8LineLidarIntrinsics
DualAxisLidarIntrinsics = _par.DualAxisLidarIntrinsics
IndoorLidar, DualAxisLidar = _gen.IndoorLidar, _gen.DualAxisLidar
create_lidar, get_lidar_type = _gen.create_lidar, _gen.get_lidar_type

__all__ = ["LidarIntrinsics", "Indoor8LineLidarIntrinsics", "DualAxisLidarIntrinsics", "IndoorLidar",
           "DualAxisLidar", "create_lidar", "get_lidar_type"]

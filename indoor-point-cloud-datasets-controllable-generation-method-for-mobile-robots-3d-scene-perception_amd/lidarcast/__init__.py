"""lidarcast -- Python face of liblidarcast, the MI355X-native LiDAR ray-cast scan engine."""
from ._capi import LIB_PATH, LRC_INVALID_PRIM, LidarcastError, load
from .core import (ATTRS, FRAME_ATTRS, Context, DeviceHits, DirectionTable, NearestIndex, OccupancyIndex, PinnedPool, ScanPipe, Scene,
                   bake_triangle_labels)

__all__ = ["LIB_PATH", "LRC_INVALID_PRIM", "LidarcastError", "load", "ATTRS", "Context",
           "DeviceHits", "DirectionTable", "Scene", "ScanPipe", "PinnedPool", "FRAME_ATTRS", "NearestIndex", "OccupancyIndex", "bake_triangle_labels", "version", "device_count"]


def version():
    return load().lrc_version().decode()


def device_count():
    return int(load().lrc_device_count())

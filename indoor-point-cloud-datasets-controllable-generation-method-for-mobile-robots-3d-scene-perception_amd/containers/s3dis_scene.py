"""Room mesh + bounds (reference: containers/s3dis_scene.py:16-46 RoomBounds, :117-210 S3DISScene).
Only what the scan loop reads: the mesh, the bounds and the bounding-box volume."""
from dataclasses import dataclass

import numpy as np


@dataclass
class RoomBounds:
    x_min: float
    x_max: float
    y_min: float
    y_max: float
    z_min: float
    z_max: float

    def get_volume(self) -> float:
        return (self.x_max - self.x_min) * (self.y_max - self.y_min) * (self.z_max - self.z_min)

    def get_center(self) -> np.ndarray:
        return np.array([(self.x_min + self.x_max) / 2, (self.y_min + self.y_max) / 2,
                         (self.z_min + self.z_max) / 2])

    @classmethod
    def from_vertices(cls, vertices) -> "RoomBounds":
        v = np.asarray(vertices)
        lo, hi = v.min(axis=0), v.max(axis=0)
        return cls(float(lo[0]), float(hi[0]), float(lo[1]), float(hi[1]), float(lo[2]), float(hi[2]))


class S3DISScene:
    def __init__(self, scene_name, room_mesh, room_bounds: RoomBounds = None):
        self.scene_name = scene_name
        self.room_mesh = room_mesh
        self.room_bounds = room_bounds or RoomBounds.from_vertices(room_mesh.vertices)

"""Room mesh + bounds + free-form semantic notes (reference: containers/s3dis_scene.py:13-218).

Same names and results as the reference's RoomBounds / SemanticInfo / S3DISScene; mesh files go through this package's
PLY reader / writer (lidarcast.ply) instead of Open3D, and the mesh is anything with ``.vertices`` / ``.triangles``.
"""
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np


class RoomBounds:
    """Axis-aligned box of a room."""

    def __init__(self, x_min: float, x_max: float, y_min: float, y_max: float, z_min: float, z_max: float):
        self.x_min, self.x_max = x_min, x_max
        self.y_min, self.y_max = y_min, y_max
        self.z_min, self.z_max = z_min, z_max

    def __eq__(self, other):
        return isinstance(other, RoomBounds) and self.to_dict() == other.to_dict()

    def __repr__(self) -> str:
        return "RoomBounds(" + ", ".join(f"{k}={v}" for k, v in self.to_dict().items()) + ")"

    def get_center(self) -> np.ndarray:
        return np.array([(self.x_min + self.x_max) / 2, (self.y_min + self.y_max) / 2,
                         (self.z_min + self.z_max) / 2])

    def get_size(self) -> np.ndarray:
        return np.array([self.x_max - self.x_min, self.y_max - self.y_min, self.z_max - self.z_min])

    def get_volume(self) -> float:
        size = self.get_size()
        return size[0] * size[1] * size[2]

    def is_point_inside(self, point: np.ndarray) -> bool:
        return (self.x_min <= point[0] <= self.x_max and self.y_min <= point[1] <= self.y_max
                and self.z_min <= point[2] <= self.z_max)

    def to_dict(self) -> Dict[str, float]:
        return {"x_min": self.x_min, "x_max": self.x_max, "y_min": self.y_min, "y_max": self.y_max,
                "z_min": self.z_min, "z_max": self.z_max}

    @classmethod
    def from_dict(cls, bounds_dict: Dict[str, float]) -> "RoomBounds":
        return cls(**bounds_dict)

    @classmethod
    def from_mesh(cls, mesh) -> "RoomBounds":
        """Bounds of the mesh vertices (numpy scalars, as the reference returns them)."""
        v = np.asarray(mesh.vertices)
        return cls(x_min=v[:, 0].min(), x_max=v[:, 0].max(), y_min=v[:, 1].min(), y_max=v[:, 1].max(),
                   z_min=v[:, 2].min(), z_max=v[:, 2].max())

    @classmethod
    def from_vertices(cls, vertices) -> "RoomBounds":
        """Bounds of a vertex array as Python floats (what the simulator passes on, s3dis_simulator.py:93-101)."""
        v = np.asarray(vertices)
        lo, hi = v.min(axis=0), v.max(axis=0)
        return cls(float(lo[0]), float(hi[0]), float(lo[1]), float(hi[1]), float(lo[2]), float(hi[2]))


class SemanticInfo:
    """Room type, furniture notes and a name -> id label table."""

    def __init__(self, room_type: str = "unknown", furniture_info: Optional[Dict[str, Any]] = None,
                 semantic_labels: Optional[Dict[str, int]] = None):
        self.room_type = room_type
        self.furniture_info = furniture_info or {}
        self.semantic_labels = semantic_labels or {}

    def add_furniture(self, name: str, position: np.ndarray, size: np.ndarray, category: str = "unknown"):
        self.furniture_info[name] = {"position": position.tolist(), "size": size.tolist(), "category": category}

    def get_furniture_count(self) -> int:
        return len(self.furniture_info)

    def to_dict(self) -> Dict[str, Any]:
        return {"room_type": self.room_type, "furniture_info": self.furniture_info,
                "semantic_labels": self.semantic_labels}


class S3DISScene:
    """A room: name, mesh, bounds (from the mesh unless given), semantic notes, mesh statistics."""

    def __init__(self, scene_name: str, room_mesh, room_bounds: Optional[RoomBounds] = None,
                 semantic_info: Optional[SemanticInfo] = None):
        self.scene_name = scene_name
        self.room_mesh = room_mesh
        self.room_bounds = room_bounds if room_bounds is not None else RoomBounds.from_mesh(room_mesh)
        self.semantic_info = semantic_info if semantic_info is not None else SemanticInfo()
        self._refresh_statistics()

    def _refresh_statistics(self):
        self.num_vertices = len(self.room_mesh.vertices)
        self.num_triangles = len(self.room_mesh.triangles)
        self.mesh_volume = self._calculate_mesh_volume()

    def _calculate_mesh_volume(self) -> float:
        """Bounding-box volume stands in for the mesh volume, as in the reference."""
        return self.room_bounds.get_volume()

    def get_bounds_center(self) -> np.ndarray:
        return self.room_bounds.get_center()

    def get_bounds_size(self) -> np.ndarray:
        return self.room_bounds.get_size()

    def is_point_inside(self, point: np.ndarray) -> bool:
        return self.room_bounds.is_point_inside(point)

    def get_mesh_statistics(self) -> Dict[str, Any]:
        return {"num_vertices": self.num_vertices, "num_triangles": self.num_triangles,
                "volume": self.mesh_volume, "bounds": self.room_bounds.to_dict()}

    def save_mesh(self, output_path: Path):
        from lidarcast.ply import write_triangle_mesh
        output_path = Path(output_path)
        output_path.parent.mkdir(parents=True, exist_ok=True)
        write_triangle_mesh(output_path, self.room_mesh)

    def load_mesh(self, mesh_path: Path) -> bool:
        """Replace the mesh by the file's; False (scene unchanged apart from an empty mesh) if it cannot be read."""
        try:
            from lidarcast.ply import read_triangle_mesh
            self.room_mesh = read_triangle_mesh(mesh_path)
            if len(self.room_mesh.vertices) == 0:
                return False
            self.room_bounds = RoomBounds.from_mesh(self.room_mesh)
            self._refresh_statistics()
            return True
        except Exception:                                          # noqa: BLE001 - the reference swallows everything
            return False

    def to_dict(self) -> Dict[str, Any]:
        return {"scene_name": self.scene_name, "room_bounds": self.room_bounds.to_dict(),
                "semantic_info": self.semantic_info.to_dict(), "mesh_statistics": self.get_mesh_statistics()}

    @classmethod
    def from_mesh_file(cls, scene_name: str, mesh_path: Path,
                       semantic_info: Optional[SemanticInfo] = None) -> "S3DISScene":
        from lidarcast.ply import read_triangle_mesh
        mesh = read_triangle_mesh(mesh_path)
        if len(mesh.vertices) == 0:
            raise ValueError(f"Cannot load mesh file: {mesh_path}")
        return cls(scene_name, mesh, semantic_info=semantic_info)

    def __repr__(self) -> str:
        return (f"S3DISScene(name='{self.scene_name}', vertices={self.num_vertices}, "
                f"triangles={self.num_triangles}, bounds={self.room_bounds.get_size()})")

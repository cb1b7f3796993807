"""The room a scan runs in: mesh, axis-aligned bounds, free-form semantic notes.

Public surface of the reference's ``containers/s3dis_scene.py`` (RoomBounds, SemanticInfo, S3DISScene) with the same
return values (tests/test_containers.py::test_scene_record_surface).  Two things differ by design: mesh files are
read and written by this package's own PLY code (``lidarcast.ply``) because Open3D is optional here, and a "mesh" is
any object with ``.vertices`` (V,3) and ``.triangles`` (T,3) -- an Open3D legacy TriangleMesh qualifies.
"""
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np

_AXES = ("x", "y", "z")


class RoomBounds:
    """Axis-aligned box ``[x_min, x_max] x [y_min, y_max] x [z_min, z_max]`` of a room."""

    def __init__(self, x_min: float, x_max: float, y_min: float, y_max: float, z_min: float, z_max: float):
        self.x_min, self.x_max = x_min, x_max
        self.y_min, self.y_max = y_min, y_max
        self.z_min, self.z_max = z_min, z_max

    # construction ------------------------------------------------------------------------------------------
    @staticmethod
    def _column_bounds(v):
        """``v.min(axis=0), v.max(axis=0)`` of a (V,3) array, column by column: the same values (a minimum is exact) as
        numpy scalars of the same type, seven times faster -- numpy reduces an axis-0 minimum of a C-ordered (V,3) array three
        elements at a time (8 ms for 300 k vertices: two thirds of a six-scene batch went there)."""
        if v.ndim != 2 or v.shape[0] == 0:
            return v.min(axis=0), v.max(axis=0)
        return [v[:, k].min() for k in range(v.shape[1])], [v[:, k].max() for k in range(v.shape[1])]

    @classmethod
    def from_vertices(cls, vertices) -> "RoomBounds":
        """Bounds of a (V,3) array as Python floats -- what the simulator hands on (s3dis_simulator.py:93-101)."""
        lo, hi = cls._column_bounds(np.asarray(vertices))
        return cls(float(lo[0]), float(hi[0]), float(lo[1]), float(hi[1]), float(lo[2]), float(hi[2]))

    @classmethod
    def from_mesh(cls, mesh) -> "RoomBounds":
        """Bounds of the mesh vertices, left as numpy scalars like the reference's."""
        lo, hi = cls._column_bounds(np.asarray(mesh.vertices))
        return cls(x_min=lo[0], x_max=hi[0], y_min=lo[1], y_max=hi[1], z_min=lo[2], z_max=hi[2])

    @classmethod
    def from_dict(cls, bounds_dict: Dict[str, float]) -> "RoomBounds":
        return cls(**bounds_dict)

    def to_dict(self) -> Dict[str, float]:
        return {f"{a}_{end}": getattr(self, f"{a}_{end}") for a in _AXES for end in ("min", "max")}

    # geometry ----------------------------------------------------------------------------------------------
    def _lo(self) -> np.ndarray:
        return np.array([self.x_min, self.y_min, self.z_min])

    def _hi(self) -> np.ndarray:
        return np.array([self.x_max, self.y_max, self.z_max])

    def get_size(self) -> np.ndarray:
        return np.array([self.x_max - self.x_min, self.y_max - self.y_min, self.z_max - self.z_min])

    def get_center(self) -> np.ndarray:
        return np.array([(self.x_min + self.x_max) / 2, (self.y_min + self.y_max) / 2,
                         (self.z_min + self.z_max) / 2])

    def get_volume(self) -> float:
        dx, dy, dz = self.get_size()
        return dx * dy * dz

    def is_point_inside(self, point: np.ndarray) -> bool:
        """Closed box test on the first three coordinates of point."""
        return all(lo <= c <= hi for lo, c, hi in zip(self._lo(), point, self._hi()))

    def __eq__(self, other):
        return isinstance(other, RoomBounds) and self.to_dict() == other.to_dict()

    def __repr__(self) -> str:
        return "RoomBounds(" + ", ".join(f"{k}={v}" for k, v in self.to_dict().items()) + ")"


class SemanticInfo:
    """Room type, a furniture notebook (name -> position / size / category) and a label-name -> id table."""

    def __init__(self, room_type: str = "unknown", furniture_info: Optional[Dict[str, Any]] = None,
                 semantic_labels: Optional[Dict[str, int]] = None):
        self.room_type = room_type
        self.furniture_info = furniture_info or {}
        self.semantic_labels = semantic_labels or {}

    def get_furniture_count(self) -> int:
        return len(self.furniture_info)

    def add_furniture(self, name: str, position: np.ndarray, size: np.ndarray, category: str = "unknown"):
        """Note a piece of furniture; arrays are stored as plain lists so the record stays JSON-ready."""
        self.furniture_info[name] = dict(position=position.tolist(), size=size.tolist(), category=category)

    def to_dict(self) -> Dict[str, Any]:
        return dict(room_type=self.room_type, furniture_info=self.furniture_info,
                    semantic_labels=self.semantic_labels)


class S3DISScene:
    """A room: name, mesh, bounds (taken from the mesh unless given), semantic notes, mesh statistics.

    ``mesh_volume`` is the bounding-box volume, the reference's stand-in for the enclosed volume; it is what the scan
    density of a frame is normalised by (s3dis_simulator.py:281)."""

    def __init__(self, scene_name: str, room_mesh, room_bounds: Optional[RoomBounds] = None,
                 semantic_info: Optional[SemanticInfo] = None):
        self.scene_name = scene_name
        self.room_mesh = room_mesh
        self.semantic_info = SemanticInfo() if semantic_info is None else semantic_info
        self.room_bounds = RoomBounds.from_mesh(room_mesh) if room_bounds is None else room_bounds
        self._count_mesh()

    def _count_mesh(self):
        self.num_vertices, self.num_triangles = len(self.room_mesh.vertices), len(self.room_mesh.triangles)
        self.mesh_volume = self._calculate_mesh_volume()

    def _calculate_mesh_volume(self) -> float:
        return self.room_bounds.get_volume()

    # files -------------------------------------------------------------------------------------------------
    @classmethod
    def from_mesh_file(cls, scene_name: str, mesh_path: Path,
                       semantic_info: Optional[SemanticInfo] = None) -> "S3DISScene":
        from lidarcast.ply import read_triangle_mesh
        mesh = read_triangle_mesh(mesh_path)
        if len(mesh.vertices) == 0:
            raise ValueError(f"Cannot load mesh file: {mesh_path}")
        return cls(scene_name, mesh, semantic_info=semantic_info)

    def load_mesh(self, mesh_path: Path) -> bool:
        """Swap in the mesh of a file and refresh bounds and statistics; False if it cannot be read or is empty
        (every exception is swallowed, as the reference does)."""
        try:
            from lidarcast.ply import read_triangle_mesh
            self.room_mesh = read_triangle_mesh(mesh_path)
            if len(self.room_mesh.vertices) == 0:
                return False
            self.room_bounds = RoomBounds.from_mesh(self.room_mesh)
            self._count_mesh()
            return True
        except Exception:                                          # noqa: BLE001
            return False

    def save_mesh(self, output_path: Path):
        from lidarcast.ply import write_triangle_mesh
        target = Path(output_path)
        target.parent.mkdir(parents=True, exist_ok=True)
        write_triangle_mesh(target, self.room_mesh)

    # queries -----------------------------------------------------------------------------------------------
    def is_point_inside(self, point: np.ndarray) -> bool:
        return self.room_bounds.is_point_inside(point)

    def get_bounds_size(self) -> np.ndarray:
        return self.room_bounds.get_size()

    def get_bounds_center(self) -> np.ndarray:
        return self.room_bounds.get_center()

    def get_mesh_statistics(self) -> Dict[str, Any]:
        return dict(num_vertices=self.num_vertices, num_triangles=self.num_triangles, volume=self.mesh_volume,
                    bounds=self.room_bounds.to_dict())

    def to_dict(self) -> Dict[str, Any]:
        return dict(scene_name=self.scene_name, room_bounds=self.room_bounds.to_dict(),
                    semantic_info=self.semantic_info.to_dict(), mesh_statistics=self.get_mesh_statistics())

    def __repr__(self) -> str:
        return (f"S3DISScene(name='{self.scene_name}', vertices={self.num_vertices}, "
                f"triangles={self.num_triangles}, bounds={self.room_bounds.get_size()})")

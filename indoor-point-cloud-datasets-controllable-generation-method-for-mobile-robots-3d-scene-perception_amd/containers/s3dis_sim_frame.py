"""Per-pose scan record the engine output is poured into (reference: containers/s3dis_sim_frame.py:12-40, :84-101).

The quality record, the incident-angle record, and the frame with its length check, accessors, filters and
dictionary round trip -- same names and results as the reference (tests/golden/make_containers_golden.py runs the
reference's own classes).  Per-ray hit attributes the HIP engine writes back (semantic / instance label per point)
ride along as optional arrays and follow the points through the filters.
"""
from dataclasses import asdict, dataclass
from typing import Any, Dict, Optional

import numpy as np


@dataclass
class ScanQuality:
    coverage_ratio: float
    num_points: int
    incident_angle_mean: float
    incident_angle_std: float
    scan_density: float
    range_mean: float
    range_std: float

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    @classmethod
    def from_dict(cls, quality_dict: Dict[str, Any]) -> "ScanQuality":
        return cls(**quality_dict)


@dataclass
class IncidentAngles:
    """Incident angles of a scan with the optional surface normals / ray directions they came from
    (reference: containers/s3dis_sim_frame.py:43-82)."""
    angles: np.ndarray
    surface_normals: Optional[np.ndarray] = None
    ray_directions: Optional[np.ndarray] = None

    def get_mean_angle(self) -> float:
        return np.mean(self.angles)

    def get_std_angle(self) -> float:
        return np.std(self.angles)

    def get_angle_distribution(self, num_bins: int = 20) -> tuple:
        return np.histogram(self.angles, bins=num_bins)

    def to_dict(self) -> Dict[str, Any]:
        def lst(a):
            return a.tolist() if a is not None else None
        return {"angles": self.angles.tolist(), "surface_normals": lst(self.surface_normals),
                "ray_directions": lst(self.ray_directions)}

    @classmethod
    def from_dict(cls, angles_dict: Dict[str, Any]) -> "IncidentAngles":
        def arr(key):      # an empty list counts as absent, as in the reference
            return np.array(angles_dict[key]) if angles_dict[key] else None
        return cls(angles=np.array(angles_dict["angles"]), surface_normals=arr("surface_normals"),
                   ray_directions=arr("ray_directions"))


class S3DISSimFrame:
    """One simulated scan: points (K,3), incident angles (K,), quality metrics."""

    def __init__(self, frame_index: int, points: np.ndarray, incident_angles: np.ndarray,
                 scan_quality: ScanQuality, frame_metadata: Optional[Dict[str, Any]] = None,
                 semantic_labels: Optional[np.ndarray] = None,
                 instance_labels: Optional[np.ndarray] = None, label_source=None):
        self.frame_index = frame_index
        self.points = points
        self.incident_angles = incident_angles
        self.scan_quality = scan_quality
        self.frame_metadata = frame_metadata or {}
        # The reference's frame carries no labels (containers/s3dis_sim_frame.py:90-101); this one offers the hit triangles'
        # as an extra.  They are either handed in, or fetched on first access from ``label_source`` (an object whose
        # ``frame_labels(i)`` returns (semantic, instance) of frame i): S3DISSimulator.run_simulation does not move four
        # of every sixteen bytes across PCIe for a caller who never looks at them.
        self._semantic_labels = semantic_labels
        self._instance_labels = instance_labels
        self._label_source = label_source
        if len(points) != len(incident_angles):
            raise ValueError(f"Point cloud count ({len(points)}) does not match incident angle count "
                             f"({len(incident_angles)})")
        for lab in (semantic_labels, instance_labels):
            if lab is not None and len(lab) != len(points):
                raise ValueError("label count does not match point cloud count")

    def _fetch_labels(self):
        src, self._label_source = self._label_source, None
        if src is not None:
            self._semantic_labels, self._instance_labels = src.frame_labels(self.frame_index)

    @property
    def semantic_labels(self):
        if self._label_source is not None:
            self._fetch_labels()
        return self._semantic_labels

    @semantic_labels.setter
    def semantic_labels(self, value):
        if self._label_source is not None:
            self._fetch_labels()
        self._semantic_labels = value

    @property
    def instance_labels(self):
        if self._label_source is not None:
            self._fetch_labels()
        return self._instance_labels

    @instance_labels.setter
    def instance_labels(self, value):
        if self._label_source is not None:
            self._fetch_labels()
        self._instance_labels = value

    def get_num_points(self) -> int:
        return len(self.points)

    def get_coverage_ratio(self) -> float:
        return self.scan_quality.coverage_ratio

    def get_scan_density(self) -> float:
        return self.scan_quality.scan_density

    def get_mean_incident_angle(self) -> float:
        return self.scan_quality.incident_angle_mean

    def get_incident_angle_std(self) -> float:
        return self.scan_quality.incident_angle_std

    def get_mean_range(self) -> float:
        return self.scan_quality.range_mean

    def get_range_std(self) -> float:
        return self.scan_quality.range_std

    def get_point_cloud_bounds(self) -> Dict[str, float]:
        if len(self.points) == 0:
            return {k: 0 for k in ("x_min", "x_max", "y_min", "y_max", "z_min", "z_max")}
        lo, hi = [self.points[:, k].min() for k in range(3)], [self.points[:, k].max() for k in range(3)]     # = min / max(axis=0)
        return {"x_min": float(lo[0]), "x_max": float(hi[0]), "y_min": float(lo[1]),
                "y_max": float(hi[1]), "z_min": float(lo[2]), "z_max": float(hi[2])}

    def get_point_cloud_center(self) -> np.ndarray:
        return np.mean(self.points, axis=0) if len(self.points) else np.array([0, 0, 0])

    def get_point_cloud_std(self) -> np.ndarray:
        return np.std(self.points, axis=0) if len(self.points) else np.array([0, 0, 0])

    def _subset(self, mask) -> "S3DISSimFrame":
        """The frame restricted to mask, quality re-derived as the reference's two filters do (:157-205): coverage
        and density scale with the kept fraction, angle statistics from the kept angles, ranges from the WORLD
        origin; an empty source frame divides by zero there and here."""
        pts, ang = self.points[mask], self.incident_angles[mask]
        frac = len(pts) / len(self.points)
        rng = np.linalg.norm(pts, axis=1) if len(pts) > 0 else None
        q = ScanQuality(
            coverage_ratio=self.scan_quality.coverage_ratio * frac, num_points=len(pts),
            incident_angle_mean=np.mean(ang) if len(ang) > 0 else 0,
            incident_angle_std=np.std(ang) if len(ang) > 0 else 0,
            scan_density=self.scan_quality.scan_density * frac,
            range_mean=np.mean(rng) if rng is not None else 0, range_std=np.std(rng) if rng is not None else 0)
        return S3DISSimFrame(frame_index=self.frame_index, points=pts, incident_angles=ang, scan_quality=q,
                             frame_metadata=self.frame_metadata.copy(),
                             semantic_labels=None if self.semantic_labels is None else self.semantic_labels[mask],
                             instance_labels=None if self.instance_labels is None else self.instance_labels[mask])

    def filter_points_by_angle(self, min_angle: float = 0, max_angle: float = np.pi / 2) -> "S3DISSimFrame":
        """Points whose incident angle lies in [min_angle, max_angle] (bounds in the unit the angles are stored in)."""
        return self._subset((self.incident_angles >= min_angle) & (self.incident_angles <= max_angle))

    def filter_points_by_range(self, min_range: float = 0, max_range: float = float("inf")) -> "S3DISSimFrame":
        """Points whose distance from the world origin lies in [min_range, max_range]."""
        r = np.linalg.norm(self.points, axis=1)
        return self._subset((r >= min_range) & (r <= max_range))

    def to_dict(self) -> Dict[str, Any]:
        return {"frame_index": self.frame_index, "points": self.points.tolist(),
                "incident_angles": self.incident_angles.tolist(),
                "scan_quality": self.scan_quality.to_dict(), "frame_metadata": self.frame_metadata}

    @classmethod
    def from_dict(cls, frame_dict: Dict[str, Any]) -> "S3DISSimFrame":
        return cls(frame_index=frame_dict["frame_index"], points=np.array(frame_dict["points"]),
                   incident_angles=np.array(frame_dict["incident_angles"]),
                   scan_quality=ScanQuality.from_dict(frame_dict["scan_quality"]),
                   frame_metadata=frame_dict.get("frame_metadata", {}))

    def __repr__(self) -> str:
        return (f"S3DISSimFrame(index={self.frame_index}, points={self.get_num_points()}, "
                f"coverage={self.get_coverage_ratio():.3f}, "
                f"mean_angle={self.get_mean_incident_angle():.3f})")

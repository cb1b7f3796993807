"""Per-pose scan record the engine output is poured into (reference: containers/s3dis_sim_frame.py:12-40, :84-101).

Only the parts the scan path touches: the quality record, the frame with its length check, and the
accessors the scene statistics read.  Per-ray hit attributes the HIP engine writes back (semantic /
instance label per point) ride along as optional arrays.
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


class S3DISSimFrame:
    """One simulated scan: points (K,3), incident angles (K,), quality metrics."""

    def __init__(self, frame_index: int, points: np.ndarray, incident_angles: np.ndarray,
                 scan_quality: ScanQuality, frame_metadata: Optional[Dict[str, Any]] = None,
                 semantic_labels: Optional[np.ndarray] = None,
                 instance_labels: Optional[np.ndarray] = None):
        self.frame_index = frame_index
        self.points = points
        self.incident_angles = incident_angles
        self.scan_quality = scan_quality
        self.frame_metadata = frame_metadata or {}
        self.semantic_labels = semantic_labels
        self.instance_labels = instance_labels
        if len(points) != len(incident_angles):
            raise ValueError(f"Point cloud count ({len(points)}) does not match incident angle count "
                             f"({len(incident_angles)})")
        for lab in (semantic_labels, instance_labels):
            if lab is not None and len(lab) != len(points):
                raise ValueError("label count does not match point cloud count")

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
        lo, hi = self.points.min(axis=0), self.points.max(axis=0)
        return {"x_min": float(lo[0]), "x_max": float(hi[0]), "y_min": float(lo[1]),
                "y_max": float(hi[1]), "z_min": float(lo[2]), "z_max": float(hi[2])}

    def to_dict(self) -> Dict[str, Any]:
        return {"frame_index": self.frame_index, "points": self.points.tolist(),
                "incident_angles": self.incident_angles.tolist(),
                "scan_quality": self.scan_quality.to_dict(), "frame_metadata": self.frame_metadata}

    def __repr__(self) -> str:
        return (f"S3DISSimFrame(index={self.frame_index}, points={self.get_num_points()}, "
                f"coverage={self.get_coverage_ratio():.3f}, "
                f"mean_angle={self.get_mean_incident_angle():.3f})")

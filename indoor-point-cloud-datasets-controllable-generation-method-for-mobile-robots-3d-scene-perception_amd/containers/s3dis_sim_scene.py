"""Scene assembly: frames -> statistics -> combined cloud -> PLY
(reference: containers/s3dis_sim_scene.py:150, :228-247, :306-377, :614-641).

Differences from the reference, all on purpose:
  * labels come from the hit triangles (written back by the trace kernel) when the frames carry
    them, instead of a 1-NN query against the raw annotated cloud at export time;
    without labels the reference's defaults apply (grey 0.5 -> 127, labels 0, :575-584);
  * the labelled PLY is written with one structured-array ``tofile`` instead of a per-point
    ``struct.pack`` loop; header and record layout are byte-identical (:619-641).
"""
from dataclasses import asdict, dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np

from .s3dis_sim_frame import S3DISSimFrame


@dataclass
class SimulationStats:
    total_frames: int
    total_points: int
    average_coverage: float
    average_scan_density: float
    average_incident_angle: float
    average_range: float
    simulation_time: float
    frames_per_second: float

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)


PLY_HEADER = (b"ply\nformat binary_little_endian 1.0\nelement vertex %d\n"
              b"property float x\nproperty float y\nproperty float z\n"
              b"property uchar red\nproperty uchar green\nproperty uchar blue\n"
              b"property ushort sem\nproperty ushort ins\nend_header\n")
PLY_RECORD = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"),
                       ("blue", "u1"), ("sem", "<u2"), ("ins", "<u2")])   # 19 bytes, packed


def write_labeled_ply(path, points, colors_u8, semantic_labels, instance_labels):
    """8-attribute binary PLY of the reference (float x,y,z; uchar r,g,b; ushort sem, ins)."""
    n = len(points)
    rec = np.empty(n, dtype=PLY_RECORD)
    rec["x"], rec["y"], rec["z"] = points[:, 0], points[:, 1], points[:, 2]
    rec["red"], rec["green"], rec["blue"] = colors_u8[:, 0], colors_u8[:, 1], colors_u8[:, 2]
    rec["sem"], rec["ins"] = semantic_labels, instance_labels
    with open(path, "wb") as f:
        f.write(PLY_HEADER % n)
        rec.tofile(f)


def read_labeled_ply(path):
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline()
            if line.startswith(b"element vertex"):
                n = int(line.split()[-1])
            if line.strip() == b"end_header":
                break
        return np.fromfile(f, dtype=PLY_RECORD, count=n)


class S3DISSimScene:
    """All frames of one simulated scene."""

    def __init__(self, scene_name: str, simulation_config: Optional[Dict[str, Any]] = None,
                 mesh: Optional[object] = None, s3dis_data_root: Optional[str] = None,
                 area: Optional[str] = None, room: Optional[str] = None):
        self.scene_name = scene_name
        self.simulation_config = simulation_config or {}
        self.frames: List[S3DISSimFrame] = []
        self.statistics: Optional[SimulationStats] = None
        self.mesh = mesh
        self.s3dis_data_root, self.area, self.room = s3dis_data_root, area, room
        self._annotated = None        # (points f64 (M,3), colors (M,3) in 0..1, sem (M,), ins (M,))
        self._nn = None

    def append_frame(self, frame: S3DISSimFrame):
        self.frames.append(frame)

    def get_total_frames(self) -> int:
        return len(self.frames)

    def get_total_points(self) -> int:
        return sum(f.get_num_points() for f in self.frames)

    def _mean(self, getter) -> float:
        return float(np.mean([getter(f) for f in self.frames])) if self.frames else 0.0

    def get_average_coverage(self) -> float:
        return self._mean(S3DISSimFrame.get_coverage_ratio)

    def get_average_scan_density(self) -> float:
        return self._mean(S3DISSimFrame.get_scan_density)

    def get_average_incident_angle(self) -> float:
        return self._mean(S3DISSimFrame.get_mean_incident_angle)

    def get_average_range(self) -> float:
        return self._mean(S3DISSimFrame.get_mean_range)

    def compute_statistics(self, simulation_time: float = 0.0):
        n = self.get_total_frames()
        self.statistics = SimulationStats(
            total_frames=n, total_points=self.get_total_points(),
            average_coverage=self.get_average_coverage(),
            average_scan_density=self.get_average_scan_density(),
            average_incident_angle=self.get_average_incident_angle(),
            average_range=self.get_average_range(),
            simulation_time=simulation_time if n else 0.0,
            frames_per_second=n / simulation_time if (n and simulation_time > 0) else 0.0)

    # ---- assembly ---------------------------------------------------------------------------------
    def combined_points(self) -> np.ndarray:
        """np.vstack of the non-empty frames in frame order (reference :326, :362)."""
        parts = [f.points for f in self.frames if len(f.points) > 0]
        return np.vstack(parts) if parts else np.empty((0, 3), dtype=np.float32)

    def combined_labels(self):
        sem, ins = [], []
        for f in self.frames:
            k = len(f.points)
            if k == 0:
                continue
            sem.append(f.semantic_labels if f.semantic_labels is not None else np.zeros(k, np.uint16))
            ins.append(f.instance_labels if f.instance_labels is not None else np.zeros(k, np.uint16))
        if not sem:
            return np.empty(0, np.uint16), np.empty(0, np.uint16)
        return np.concatenate(sem).astype(np.uint16), np.concatenate(ins).astype(np.uint16)

    # ---- export-time labels from an annotated cloud (the reference's semantics) -------------------------
    def set_annotated_cloud(self, points, colors, semantic_labels, instance_labels):
        """The arrays the reference reads from the S3DIS annotations and caches
        (containers/s3dis_sim_scene.py:396-409).  Loading the annotation files is out of scope here."""
        self._annotated = (np.ascontiguousarray(points, dtype=np.float64), np.asarray(colors),
                           np.asarray(semantic_labels), np.asarray(instance_labels))
        if self._nn is not None:
            self._nn.close()
        self._nn = None

    def _get_default_colors_and_labels(self, num_points: int):
        return (np.ones((num_points, 3), dtype=np.float32) * 0.5, np.zeros(num_points, dtype=np.uint16),
                np.zeros(num_points, dtype=np.uint16))

    def _get_colors_and_labels_from_s3dis(self, points: np.ndarray):
        """Nearest annotated point per hit point -> (colors, semantic, instance); defaults without a cloud.
        Same contract as the reference method of this name (:379-427); the ball-tree query is the GPU 1-NN."""
        if self._annotated is None or len(points) == 0:
            return self._get_default_colors_and_labels(len(points))
        if self._nn is None:
            import lidarcast
            self._nn_ctx = lidarcast.Context(0)
            self._nn = lidarcast.NearestIndex(self._nn_ctx, self._annotated[0])
        idx = self._nn.query(points)
        _, colors, sem, ins = self._annotated
        return colors[idx], sem[idx], ins[idx]

    def save_results(self, output_dir, formats=("txt",)):
        out = Path(output_dir)
        out.mkdir(parents=True, exist_ok=True)
        if self.statistics is None:
            self.compute_statistics()
        if "txt" in formats:
            with open(out / "simulation_statistics.txt", "w", encoding="utf-8") as f:
                for k, v in self.statistics.to_dict().items():
                    f.write(f"{k}: {v}\n")
        pts = self.combined_points()
        if len(pts) == 0:
            return
        if self._annotated is not None:       # reference path: 1-NN into the annotated cloud, per frame (:347-356)
            cols, sems, inss = [], [], []
            for f in self.frames:
                if len(f.points) > 0:
                    c, s_, i_ = self._get_colors_and_labels_from_s3dis(f.points)
                    cols.append(c); sems.append(s_); inss.append(i_)
            colors = (np.vstack(cols) * 255).astype(np.uint8)
            sem, ins = np.concatenate(sems).astype(np.uint16), np.concatenate(inss).astype(np.uint16)
        else:                                 # labels written back by the trace kernel; reference default grey
            sem, ins = self.combined_labels()
            colors = np.full((len(pts), 3), int(0.5 * 255), dtype=np.uint8)
        write_labeled_ply(out / "combined_pointcloud_with_label.ply", pts, colors, sem, ins)

"""S3DIS point-level annotations: the source of the annotated cloud the scene export labels hit points from
(reference: s3dis_annotation_loader.py:13-516; containers/s3dis_sim_scene.py:505-577 is its only caller on the scan path).

Same class and method names, label tables and return values as the reference (tests/golden/make_annotation_golden.py runs
the reference module on a synthetic annotation folder).  File layout read: ``<root>/<area>/<room>/Annotations/<class>_<k>.txt``
with ``x y z [r g b]`` rows.  Kept quirks, because a drop-in returns what the reference returns:
  * instances are numbered by the order ``glob`` lists a class's files, not by the number in the file name;
  * ``create_labeled_pointcloud`` looks the dictionary KEY up in the class-id table, so it only labels dictionaries
    keyed by plain class names (the loader's own keys are ``<class>_<k>`` and get no labels);
  * classes with a mapping but no S3DIS id (``stairs``) contribute points without labels.
Not reproduced: the reference's console output, and its two ``get_semantic_colors*`` helpers, which call a method that
does not exist.  The nearest-neighbour colour assignment runs on the GPU (lidarcast.NearestIndex).
"""
import glob
import os
from typing import Dict, Tuple

import numpy as np

_EMPTY_POINTS = np.array([]).reshape(0, 3)


class S3DISAnnotationLoader:
    """Reads a room's annotation files into {instance name: (N,3) points} and flattens them into labelled clouds."""

    def __init__(self, data_root: str):
        self.data_root = data_root
        # S3DIS class -> LiDAR-Net class; its keys are the classes that are read at all
        self.class_mapping = {"floor": "floor", "ceiling": "ceiling", "wall": "wall", "window": "window",
                              "table": "table", "chair": "chair", "sofa": "sofa", "bookcase": "bookshelf",
                              "board": "blackboard", "stairs": "stair"}
        self.valid_classes = list(self.class_mapping.keys())
        self.s3dis_class_ids = {name: i for i, name in enumerate(
            ("ceiling", "floor", "wall", "beam", "column", "window", "door", "table", "chair", "sofa", "bookcase",
             "board", "clutter"))}

    def _load_annotation_file(self, file_path: str) -> np.ndarray:
        """(N,3) float64 coordinates of one annotation file; rows that do not start with three numbers are skipped,
        an unreadable file gives an empty array."""
        rows = []
        try:
            with open(file_path, "r") as f:
                for line in f:
                    line = line.strip()
                    if not line or line.startswith("#"):
                        continue
                    tok = line.split()
                    if len(tok) < 3:
                        continue
                    try:
                        rows.append([float(tok[0]), float(tok[1]), float(tok[2])])
                    except ValueError:
                        continue
        except Exception:                                          # noqa: BLE001
            return _EMPTY_POINTS.copy()
        return np.array(rows) if rows else _EMPTY_POINTS.copy()

    def load_room_annotations(self, area: str, room: str) -> Dict[str, np.ndarray]:
        annotation_dir = os.path.join(self.data_root, area, room, "Annotations")
        if not os.path.exists(annotation_dir):
            raise FileNotFoundError(f"Annotation directory missing: {annotation_dir}")
        out = {}
        for class_name in self.valid_classes:
            for k, path in enumerate(glob.glob(os.path.join(annotation_dir, f"{class_name}_*.txt"))):
                pts = self._load_annotation_file(path)
                if len(pts) > 0:
                    out[f"{class_name}_{k + 1}"] = pts
        return out

    def create_labeled_pointcloud(self, room_annotations: Dict[str, np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
        """(points, int32 labels) of a dictionary keyed by CLASS names (see the module docstring)."""
        pts, labs = [], []
        for key, p in room_annotations.items():
            if len(p) == 0:
                continue
            pts.append(p)
            cid = self.s3dis_class_ids.get(key, -1)
            if cid >= 0:
                labs.append(np.full(len(p), cid, dtype=np.int32))
        if not pts:
            return _EMPTY_POINTS.copy(), np.array([], dtype=np.int32)
        return np.vstack(pts), np.concatenate(labs)

    def create_labeled_pointcloud_with_instances(self, room_annotations: Dict[str, np.ndarray]):
        """(points, int32 labels, int32 instances) of a dictionary keyed ``<class>_<k>``; on any failure three empty
        arrays, as the reference returns."""
        pts, labs, inss = [], [], []
        for name, p in room_annotations.items():
            if len(p) == 0:
                continue
            pts.append(p)
            cid = self.s3dis_class_ids.get(name.split("_")[0] if "_" in name else name, -1)
            if cid < 0:
                continue
            inst = 1
            if "_" in name:
                try:
                    inst = int(name.split("_")[-1])
                except ValueError:
                    inst = 1
            labs.append(np.full(len(p), cid, dtype=np.int32))
            inss.append(np.full(len(p), inst, dtype=np.int32))
        empty = (_EMPTY_POINTS.copy(), np.array([], dtype=np.int32), np.array([], dtype=np.int32))
        if not pts:
            return empty
        try:
            return np.vstack(pts), np.concatenate(labs), np.concatenate(inss)
        except Exception:                                          # noqa: BLE001
            return empty

    def filter_valid_labels(self, labels: np.ndarray) -> np.ndarray:
        """Labels of classes without a mapping become -1."""
        keep_ids = [self.s3dis_class_ids[c] for c in self.valid_classes if c in self.s3dis_class_ids]
        out = labels.copy()
        out[~np.isin(labels, keep_ids)] = -1
        return out


class S3DISColorEncoder:
    """Deterministic label (+ instance) <-> RGB code: class base colour, instance id added to the blue channel."""

    def __init__(self):
        self.class_base_colors = {"floor": [100, 50, 25], "ceiling": [200, 200, 200], "wall": [150, 150, 150],
                                  "window": [50, 150, 200], "table": [100, 50, 25], "chair": [200, 50, 50],
                                  "sofa": [150, 50, 150], "bookcase": [50, 100, 50], "board": [25, 25, 25],
                                  "stairs": [200, 150, 50]}
        self.id_to_class = {1: "floor", 0: "ceiling", 2: "wall", 5: "window", 7: "table", 8: "chair", 9: "sofa",
                            10: "bookcase", 11: "board"}
        self.max_instances_per_class = 20
        self.instance_step = 1

    def _base_table(self):
        """(ids, (len,3) float64 base colours) of the classes that have an id"""
        ids = np.array(list(self.id_to_class.keys()), dtype=np.int64)
        return ids, np.array([self.class_base_colors[self.id_to_class[int(i)]] for i in ids], dtype=np.float64)

    def encode_labels_to_colors(self, labels: np.ndarray) -> np.ndarray:
        """(N,3) float32 in [0,1]: base colour / 255, black for labels without a colour."""
        labels = np.asarray(labels)
        ids, base = self._base_table()
        colors = np.zeros((len(labels), 3), dtype=np.float32)
        for cid, rgb in zip(ids, base):
            colors[labels == cid] = [c / 255.0 for c in rgb]
        return colors

    def encode_labels_and_instances_to_colors(self, labels: np.ndarray, instances: np.ndarray) -> np.ndarray:
        """As above with min(max(instance, 0), 19) added to the blue channel before the division."""
        labels, instances = np.asarray(labels), np.asarray(instances)
        ids, base = self._base_table()
        inst = np.minimum(np.where(instances >= 0, instances.astype(np.int64), 0), self.max_instances_per_class - 1)
        colors = np.zeros((len(labels), 3), dtype=np.float32)
        for cid, rgb in zip(ids, base):
            m = labels == cid
            if m.any():
                colors[m, 0], colors[m, 1] = rgb[0] / 255.0, rgb[1] / 255.0
                colors[m, 2] = (rgb[2] + inst[m]) / 255.0
        return colors

    def decode_colors_to_labels_and_instances(self, colors: np.ndarray) -> tuple:
        """Nearest base colour in |dR| + |dG| (first class in table order on ties) gives the label; the blue surplus
        over that class's base, clamped to [0, 19], the instance.  A class without an id leaves label 0."""
        colors = np.asarray(colors)
        c255 = (colors * 255).astype(np.int32)
        names = list(self.class_base_colors.keys())
        base = np.array([self.class_base_colors[n] for n in names], dtype=np.int64)
        dist = np.abs(c255[:, None, 0].astype(np.int64) - base[None, :, 0]) + \
            np.abs(c255[:, None, 1].astype(np.int64) - base[None, :, 1])
        best = np.argmin(dist, axis=1) if len(colors) else np.zeros(0, dtype=np.int64)
        class_to_id = {}
        for cid, name in self.id_to_class.items():
            class_to_id.setdefault(name, cid)
        lab_of = np.array([class_to_id.get(n, 0) for n in names], dtype=np.int32)
        labels = lab_of[best].astype(np.int32)
        surplus = np.maximum(0, c255[:, 2].astype(np.int64) - base[best, 2])
        instances = np.minimum(surplus, self.max_instances_per_class - 1).astype(np.int32)
        return labels, instances

    def _assign_colors_to_points(self, input_points, annotation_points, annotation_labels):
        """Colour of the nearest annotated point's label (reference: sklearn ball_tree 1-NN; here the GPU 1-NN, same
        indices)."""
        import lidarcast
        nn = lidarcast.NearestIndex(lidarcast.Context(0), np.asarray(annotation_points, dtype=np.float64))
        idx = nn.query(np.asarray(input_points, dtype=np.float32))
        nn.close()
        return self.encode_labels_to_colors(np.asarray(annotation_labels)[idx])


def load_s3dis_room_labels(data_root: str, area: str, room: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(points, labels, colours) of a room through ``create_labeled_pointcloud``; three empty arrays when the folder is
    missing, holds nothing, or anything fails -- which, with the loader's instance-named keys, is what the reference
    returns for every real room (see the module docstring)."""
    empty = (_EMPTY_POINTS.copy(), np.array([], dtype=np.int32), _EMPTY_POINTS.copy())
    try:
        loader = S3DISAnnotationLoader(data_root)
        rooms = loader.load_room_annotations(area, room)
        if not rooms:
            return empty
        points, labels = loader.create_labeled_pointcloud(rooms)
        if len(points) == 0:
            return empty
        labels = loader.filter_valid_labels(labels)
        return points, labels, S3DISColorEncoder().encode_labels_to_colors(labels)
    except Exception:                                              # noqa: BLE001
        return empty

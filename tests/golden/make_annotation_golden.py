#!/usr/bin/env python3
"""Golden outputs of the reference's s3dis_annotation_loader.py on a synthetic annotation folder (build container
only; the module needs numpy + torch, both present; its prints are silenced).

    python tests/golden/make_annotation_golden.py     # writes tests/golden/annotation_golden.npz / .json
"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_annotation_loader", os.path.join(REF, "s3dis_annotation_loader.py"))
M = importlib.util.module_from_spec(spec)
with contextlib.redirect_stdout(io.StringIO()):
    spec.loader.exec_module(M)

FILES = {   # one file per class (glob order of several files of one class is file-system dependent)
    "chair_3.txt": 40, "table_1.txt": 25, "wall_7.txt": 60, "floor_1.txt": 30, "stairs_1.txt": 10, "clutter_2.txt": 15,
    "window_1.txt": 0, "board_1.txt": 5,
}


def write_room(root, rng):
    d = os.path.join(root, "Area_9", "office_1", "Annotations")
    os.makedirs(d)
    for name, n in FILES.items():
        with open(os.path.join(d, name), "w") as f:
            f.write("# header line\n\n")
            for i in range(n):
                x, y, z = rng.uniform(-5, 5, 3)
                r, g, b = rng.integers(0, 256, 3)
                f.write(f"{x:.6f} {y:.6f} {z:.6f} {r} {g} {b}\n" if i % 7 else f"{x:.6f} {y:.6f} {z:.6f}\n")
            if name == "table_1.txt":
                f.write("not a number 1 2\n1.0 2.0\n")
    return d


def main():
    A, J = {}, {}
    rng = np.random.default_rng(17)
    with tempfile.TemporaryDirectory() as root, contextlib.redirect_stdout(io.StringIO()):
        d = write_room(root, rng)
        J["annotation_files"] = {name: open(os.path.join(d, name)).read() for name in FILES}   # the input, as text
        loader = M.S3DISAnnotationLoader(root)
        J["class_mapping"], J["valid_classes"], J["s3dis_class_ids"] = loader.class_mapping, loader.valid_classes, loader.s3dis_class_ids
        rooms = loader.load_room_annotations("Area_9", "office_1")
        J["room_keys"] = list(rooms.keys())
        for k, v in rooms.items():
            A[f"room_{k}"] = v
        p, l, i = loader.create_labeled_pointcloud_with_instances(rooms)
        A["wi_points"], A["wi_labels"], A["wi_instances"] = p, l, i
        try:
            loader.create_labeled_pointcloud(rooms)
            J["plain_on_instance_keys"] = "ok"
        except Exception as e:                                     # noqa: BLE001
            J["plain_on_instance_keys"] = type(e).__name__
        by_class = {"chair": rooms["chair_1"], "wall": rooms["wall_1"], "stairs": rooms["stairs_1"][:0], "floor": rooms["floor_1"]}
        p2, l2 = loader.create_labeled_pointcloud(by_class)
        A["plain_points"], A["plain_labels"] = p2, l2
        lab = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, -1, 40], dtype=np.int32)
        A["filter_in"], A["filter_out"] = lab, loader.filter_valid_labels(lab)
        try:
            loader.load_room_annotations("Area_9", "nowhere")
        except Exception as e:                                     # noqa: BLE001
            J["missing_room"] = type(e).__name__
        e0, e1, e2 = loader.create_labeled_pointcloud_with_instances({})
        J["empty_shapes"] = [list(e0.shape), list(e1.shape), list(e2.shape), str(e1.dtype)]
        enc = M.S3DISColorEncoder()
        J["class_base_colors"], J["id_to_class"] = enc.class_base_colors, {str(k): v for k, v in enc.id_to_class.items()}
        labels = np.array([0, 1, 2, 5, 7, 8, 9, 10, 11, 3, 12, -1, 8, 8], dtype=np.int32)
        inst = np.array([0, 1, 2, 3, 4, 5, 19, 20, 45, 1, 1, 1, -3, 7], dtype=np.int32)
        A["enc_labels"], A["enc_instances"] = labels, inst
        A["enc_colors"] = enc.encode_labels_to_colors(labels)
        A["enc_colors_inst"] = enc.encode_labels_and_instances_to_colors(labels, inst)
        dl, di = enc.decode_colors_to_labels_and_instances(A["enc_colors_inst"])
        A["dec_labels"], A["dec_instances"] = dl, di
        rnd = rng.uniform(0, 1, size=(50, 3)).astype(np.float32)
        A["dec_random_in"] = rnd
        A["dec_random_labels"], A["dec_random_instances"] = enc.decode_colors_to_labels_and_instances(rnd)
        pts, labs, cols = M.load_s3dis_room_labels(root, "Area_9", "office_1")
        J["room_labels_shapes"] = [list(pts.shape), list(labs.shape), list(cols.shape)]
        pts, labs, cols = M.load_s3dis_room_labels(root, "Area_9", "nowhere")
        J["room_labels_missing_shapes"] = [list(pts.shape), list(labs.shape), list(cols.shape)]
    np.savez_compressed(os.path.join(OUT, "annotation_golden.npz"), **A)
    with open(os.path.join(OUT, "annotation_golden.json"), "w") as f:
        json.dump(J, f, indent=1)
    print("wrote", len(A), "arrays; room keys", J["room_keys"], "| plain on instance keys:", J["plain_on_instance_keys"],
          "| room labels shapes", J["room_labels_shapes"])


if __name__ == "__main__":
    main()

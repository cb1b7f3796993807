"""S3DIS annotation loader and colour codec against outputs of the reference's own module on the same synthetic
annotation folder (tests/golden/make_annotation_golden.py): tables, parsed points, labelled clouds, the quirks listed in
the module docstring, colour encode / decode."""
import json
import os

import numpy as np
import pytest

from helpers import assert_bit_equal
from s3dis_annotation_loader import S3DISAnnotationLoader, S3DISColorEncoder, load_s3dis_room_labels

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    A = np.load(os.path.join(HERE, "golden", "annotation_golden.npz"))
    with open(os.path.join(HERE, "golden", "annotation_golden.json")) as f:
        return A, json.load(f)


@pytest.fixture()
def root(tmp_path, G):
    d = tmp_path / "Area_9" / "office_1" / "Annotations"
    d.mkdir(parents=True)
    for name, text in G[1]["annotation_files"].items():
        (d / name).write_text(text)
    return str(tmp_path)


def test_loader_matches_the_reference(G, root):
    A, J = G
    loader = S3DISAnnotationLoader(root)
    assert loader.class_mapping == J["class_mapping"] and loader.valid_classes == J["valid_classes"]
    assert loader.s3dis_class_ids == J["s3dis_class_ids"]
    rooms = loader.load_room_annotations("Area_9", "office_1")
    assert list(rooms.keys()) == J["room_keys"]                  # class order of the table, instance = glob position
    for k, v in rooms.items():
        assert_bit_equal(v, A[f"room_{k}"], k)
    p, l, i = loader.create_labeled_pointcloud_with_instances(rooms)
    assert_bit_equal(p, A["wi_points"]); assert_bit_equal(l, A["wi_labels"]); assert_bit_equal(i, A["wi_instances"])
    assert len(p) > len(l)                                       # 'stairs' has a mapping but no id: points, no labels
    assert J["plain_on_instance_keys"] == "ValueError"
    with pytest.raises(ValueError):
        loader.create_labeled_pointcloud(rooms)
    by_class = {"chair": rooms["chair_1"], "wall": rooms["wall_1"], "stairs": rooms["stairs_1"][:0], "floor": rooms["floor_1"]}
    p2, l2 = loader.create_labeled_pointcloud(by_class)
    assert_bit_equal(p2, A["plain_points"]); assert_bit_equal(l2, A["plain_labels"])
    assert_bit_equal(loader.filter_valid_labels(A["filter_in"]), A["filter_out"])
    assert J["missing_room"] == "FileNotFoundError"
    with pytest.raises(FileNotFoundError):
        loader.load_room_annotations("Area_9", "nowhere")
    e = loader.create_labeled_pointcloud_with_instances({})
    assert [list(e[0].shape), list(e[1].shape), list(e[2].shape), str(e[1].dtype)] == J["empty_shapes"]
    got = load_s3dis_room_labels(root, "Area_9", "office_1")
    assert [list(a.shape) for a in got] == J["room_labels_shapes"]
    got = load_s3dis_room_labels(root, "Area_9", "nowhere")
    assert [list(a.shape) for a in got] == J["room_labels_missing_shapes"]


def test_colour_codec_matches_the_reference(G):
    A, J = G
    enc = S3DISColorEncoder()
    assert enc.class_base_colors == J["class_base_colors"]
    assert {str(k): v for k, v in enc.id_to_class.items()} == J["id_to_class"]
    assert_bit_equal(enc.encode_labels_to_colors(A["enc_labels"]), A["enc_colors"])
    assert_bit_equal(enc.encode_labels_and_instances_to_colors(A["enc_labels"], A["enc_instances"]), A["enc_colors_inst"])
    dl, di = enc.decode_colors_to_labels_and_instances(A["enc_colors_inst"])
    assert_bit_equal(dl, A["dec_labels"]); assert_bit_equal(di, A["dec_instances"])
    dl, di = enc.decode_colors_to_labels_and_instances(A["dec_random_in"])
    assert_bit_equal(dl, A["dec_random_labels"]); assert_bit_equal(di, A["dec_random_instances"])

"""CPU: frame/scene containers and the labelled PLY layout (reference containers/s3dis_sim_scene.py:614-641)."""
import struct

import numpy as np
import pytest

from containers import (S3DISSimFrame, S3DISSimScene, ScanQuality, read_labeled_ply, write_labeled_ply)


def _frame(i, k, rng):
    pts = rng.normal(size=(k, 3)).astype(np.float32)
    q = ScanQuality(k / 100, k, 0, 0, k / 10.0, float(np.mean(np.linalg.norm(pts, axis=1))) if k else 0, 0)
    return S3DISSimFrame(i, pts, np.zeros(k), q, semantic_labels=rng.integers(0, 13, k).astype(np.uint16),
                         instance_labels=rng.integers(0, 500, k).astype(np.uint16))


def test_scene_assembly_is_vstack_in_frame_order(tmp_path):
    rng = np.random.default_rng(0)
    sc = S3DISSimScene("room")
    frames = [_frame(0, 5, rng), _frame(1, 0, rng), _frame(2, 7, rng)]
    for f in frames:
        sc.append_frame(f)
    assert sc.get_total_frames() == 3 and sc.get_total_points() == 12
    assert np.array_equal(sc.combined_points(), np.vstack([frames[0].points, frames[2].points]))
    sc.compute_statistics(2.0)
    assert sc.statistics.frames_per_second == 1.5 and sc.statistics.total_points == 12
    assert abs(sc.statistics.average_coverage - np.mean([0.05, 0.0, 0.07])) < 1e-12
    sc.save_results(tmp_path)
    rec = read_labeled_ply(tmp_path / "combined_pointcloud_with_label.ply")
    assert len(rec) == 12 and (rec["red"] == 127).all()
    sem, ins = sc.combined_labels()
    assert np.array_equal(rec["sem"], sem) and np.array_equal(rec["ins"], ins)
    empty = S3DISSimScene("e")
    empty.compute_statistics(1.0)
    assert empty.statistics.total_frames == 0 and empty.statistics.frames_per_second == 0.0


def test_ply_bytes_match_the_reference_writer_layout(tmp_path):
    """Byte-for-byte what the reference's per-point struct.pack loop produces."""
    rng = np.random.default_rng(1)
    n = 17
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    col = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    sem = rng.integers(0, 13, n).astype(np.uint16)
    ins = rng.integers(0, 65535, n).astype(np.uint16)
    path = tmp_path / "a.ply"
    write_labeled_ply(path, pts, col, sem, ins)
    want = (b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
            b"property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
            b"property ushort sem\nproperty ushort ins\nend_header\n" % n)
    for i in range(n):
        want += struct.pack("<fff", *pts[i]) + struct.pack("<BBB", *col[i]) + struct.pack("<HH", sem[i], ins[i])
    assert path.read_bytes() == want


def test_mesh_ply_roundtrip(tmp_path):
    from lidarcast import synth
    from lidarcast.ply import read_triangle_mesh, write_triangle_mesh
    m = synth.unit_cube()
    write_triangle_mesh(tmp_path / "m.ply", m)
    r = read_triangle_mesh(tmp_path / "m.ply")
    assert np.array_equal(r.triangles, m.triangles) and np.allclose(r.vertices, m.vertices)


def test_scene_record_surface(tmp_path):
    """RoomBounds / SemanticInfo / S3DISScene: the reference's names and values (containers/s3dis_scene.py:13-218)."""
    from containers import RoomBounds, S3DISScene, SemanticInfo
    from lidarcast import synth
    mesh = synth.make_room(size=(4, 3, 2.5), num_boxes=1, seed=1, cell=0.5)
    b = RoomBounds.from_mesh(mesh)
    v = np.asarray(mesh.vertices)
    assert b.to_dict() == {"x_min": v[:, 0].min(), "x_max": v[:, 0].max(), "y_min": v[:, 1].min(),
                           "y_max": v[:, 1].max(), "z_min": v[:, 2].min(), "z_max": v[:, 2].max()}
    assert RoomBounds.from_dict(b.to_dict()) == b
    size = b.get_size()
    assert np.array_equal(size, v.max(axis=0) - v.min(axis=0)) and b.get_volume() == size[0] * size[1] * size[2]
    assert np.array_equal(b.get_center(), (v.max(axis=0) + v.min(axis=0)) / 2)
    assert b.is_point_inside(b.get_center()) and not b.is_point_inside(b.get_center() + size)
    info = SemanticInfo("office")
    info.add_furniture("desk", np.array([1.0, 2.0, 0.4]), np.array([1.2, 0.6, 0.8]), "table")
    assert info.get_furniture_count() == 1
    assert info.to_dict() == {"room_type": "office", "semantic_labels": {},
                              "furniture_info": {"desk": {"position": [1.0, 2.0, 0.4], "size": [1.2, 0.6, 0.8],
                                                          "category": "table"}}}
    sc = S3DISScene("room_a", mesh, semantic_info=info)
    assert sc.room_bounds == b and sc.num_vertices == len(v) and sc.num_triangles == len(mesh.triangles)
    assert sc.mesh_volume == b.get_volume() and sc.is_point_inside(sc.get_bounds_center())
    assert np.array_equal(sc.get_bounds_size(), size)
    assert sc.get_mesh_statistics() == {"num_vertices": len(v), "num_triangles": len(mesh.triangles),
                                        "volume": b.get_volume(), "bounds": b.to_dict()}
    assert sorted(sc.to_dict()) == ["mesh_statistics", "room_bounds", "scene_name", "semantic_info"]
    assert repr(sc) == f"S3DISScene(name='room_a', vertices={len(v)}, triangles={len(mesh.triangles)}, bounds={size})"
    sc.save_mesh(tmp_path / "sub" / "room_a.ply")
    sc2 = S3DISScene.from_mesh_file("again", tmp_path / "sub" / "room_a.ply")
    assert sc2.num_triangles == sc.num_triangles and sc2.semantic_info.get_furniture_count() == 0
    assert np.allclose(sc2.room_bounds.get_size(), size, atol=1e-6)       # PLY stores float32 coordinates
    assert not sc2.load_mesh(tmp_path / "missing.ply")
    assert sc2.load_mesh(tmp_path / "sub" / "room_a.ply") and sc2.num_vertices == len(v)
    (tmp_path / "empty.ply").write_text("ply\nformat ascii 1.0\nelement vertex 0\nproperty float x\nproperty float y\n"
                                        "property float z\nelement face 0\nproperty list uchar int vertex_indices\n"
                                        "end_header\n")
    with pytest.raises(ValueError):
        S3DISScene.from_mesh_file("none", tmp_path / "empty.ply")

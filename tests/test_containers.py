"""CPU: frame/scene containers and the labelled PLY layout (reference containers/s3dis_sim_scene.py:614-641)."""
import struct

import numpy as np

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

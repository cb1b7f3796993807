"""Result containers (S3DISSimFrame / S3DISSimScene / ResultExporter) against outputs of the reference's own classes
on the same seeded inputs (tests/golden/make_containers_golden.py): returned values, error behaviour, and the text /
bytes of every file they write."""
import hashlib
import json
import os

import numpy as np
import pytest

from containers import IncidentAngles, NumpyEncoder, ResultExporter, S3DISSimFrame, S3DISSimScene, ScanQuality
from helpers import assert_bit_equal

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def A():
    return np.load(os.path.join(HERE, "golden", "containers_golden.npz"))


@pytest.fixture(scope="module")
def J():
    with open(os.path.join(HERE, "golden", "containers_golden.json")) as f:
        return json.load(f)


def jsonable(o):
    return json.loads(json.dumps(o, cls=NumpyEncoder))


def frames_from(A, J):
    return [S3DISSimFrame(J[f"in_index_{k}"], A[f"in_points_{k}"], A[f"in_angles_{k}"],
                          ScanQuality(**J[f"in_quality_{k}"]),
                          {key: J[f"in_meta_{k}"][key] for key in ("waypoint", "tag")})   # the generator's key order
            for k in range(4)]


def test_frame_level(A, J):
    fr = frames_from(A, J)
    f0, empty = fr[0], fr[1]
    assert_bit_equal(f0.get_point_cloud_center(), A["center_0"])
    assert_bit_equal(f0.get_point_cloud_std(), A["std_0"])
    assert_bit_equal(empty.get_point_cloud_center(), A["center_empty"])
    assert_bit_equal(empty.get_point_cloud_std(), A["std_empty"])
    assert f0.get_point_cloud_bounds() == J["bounds_0"] and empty.get_point_cloud_bounds() == J["bounds_empty"]
    for name, g in (("angle", f0.filter_points_by_angle(20.0, 70.0)), ("angle_default", f0.filter_points_by_angle()),
                    ("range", f0.filter_points_by_range(2.0, 6.5)), ("range_none", f0.filter_points_by_range(100.0))):
        assert_bit_equal(g.points, A[f"filt_{name}_points"], name)
        assert_bit_equal(g.incident_angles, A[f"filt_{name}_angles"], name)
        assert jsonable(g.scan_quality.to_dict()) == J[f"filt_{name}_quality"], name
        assert g.frame_metadata == J[f"filt_{name}_meta"] and g.frame_metadata is not f0.frame_metadata
        assert g.frame_index == f0.frame_index
    assert J["filter_empty_frame"] == "ZeroDivisionError"
    with pytest.raises(ZeroDivisionError):
        empty.filter_points_by_range(0.0, 1.0)
    d0 = f0.to_dict()
    assert sorted(d0) == J["frame0_dict_keys"]
    back = S3DISSimFrame.from_dict(json.loads(json.dumps(d0)))
    assert_bit_equal(back.points, A["roundtrip_points_0"])          # float64 after the round trip, as in the reference
    assert_bit_equal(back.incident_angles, A["roundtrip_angles_0"])
    assert back.scan_quality.to_dict() == J["roundtrip_quality_0"]
    assert repr(f0) == J["repr_frame_0"]
    # labels written back by the engine follow the points through a filter
    lab = S3DISSimFrame(0, f0.points, f0.incident_angles, f0.scan_quality,
                        semantic_labels=np.arange(len(f0.points), dtype=np.uint16),
                        instance_labels=np.arange(len(f0.points), dtype=np.uint16)[::-1].copy())
    g = lab.filter_points_by_angle(20.0, 70.0)
    keep = (f0.incident_angles >= 20.0) & (f0.incident_angles <= 70.0)
    assert np.array_equal(g.semantic_labels, lab.semantic_labels[keep])
    assert np.array_equal(g.instance_labels, lab.instance_labels[keep])


def test_incident_angles_record(A, J):
    ia = IncidentAngles(angles=A["in_angles_0"], surface_normals=A["in_points_0"].astype(np.float64))
    hist, bins = ia.get_angle_distribution(7)
    assert_bit_equal(hist, A["ia_hist"])
    assert_bit_equal(bins, A["ia_bins"])
    assert float(ia.get_mean_angle()) == J["ia_mean"] and float(ia.get_std_angle()) == J["ia_std"]
    ia2 = IncidentAngles.from_dict(json.loads(json.dumps(ia.to_dict())))
    assert [ia2.surface_normals is not None, ia2.ray_directions is not None] == J["ia_roundtrip_has"]


def test_scene_level(A, J):
    sc = S3DISSimScene("golden_room", {"lidar": "8line", "n": 4})
    assert sc.get_frame_statistics() == J["empty_frame_statistics"]
    assert sc.get_quality_distribution() == J["empty_quality_distribution"]
    sc.compute_statistics(3.0)
    assert sc.statistics.to_dict() == J["empty_statistics"]
    for f in frames_from(A, J):
        sc.append_frame(f)
    assert jsonable(sc.get_frame_statistics()) == J["frame_statistics"]
    assert jsonable(sc.get_quality_distribution()) == J["quality_distribution"]
    sc.compute_statistics(2.5)
    assert jsonable(sc.statistics.to_dict()) == J["statistics_2p5"]
    assert repr(sc) == J["repr_scene"]
    assert [f.frame_index for f in sc.filter_frames_by_quality(0.4, 0.9).frames] == J["filter_frames_0p4_0p9"]
    for metric in ("coverage", "points", "density"):
        assert [f.frame_index for f in sc.get_best_frames(2, metric)] == J[f"best_{metric}"], metric
    assert J["best_bad_metric"] == "ValueError"
    with pytest.raises(ValueError):
        sc.get_best_frames(2, "nope")
    sd = sc.to_dict()
    assert sorted(sd) == J["scene_dict_keys"]
    sc2 = S3DISSimScene.from_dict(json.loads(json.dumps(sd, cls=NumpyEncoder)))
    assert {"frames": sc2.get_total_frames(), "points": sc2.get_total_points(),
            "statistics": jsonable(sc2.statistics.to_dict())} == J["scene_roundtrip"]


def test_written_files(A, J, tmp_path):
    fr = frames_from(A, J)
    sc = S3DISSimScene("golden_room", {"lidar": "8line", "n": 4})
    for f in fr:
        sc.append_frame(f)
    sc.compute_statistics(2.5)
    ex = ResultExporter(tmp_path / "ex")
    ex.export_statistics(sc.statistics, "txt")
    ex.export_statistics(sc.statistics, "json")
    ex.export_summary(sc, "json")
    ex.export_frames(fr[:1], "json")
    ex.export_frames(fr[2:3], "pkl")
    assert (tmp_path / "ex" / "simulation_statistics.txt").read_text() == J["file_statistics_txt"]
    assert (tmp_path / "ex" / "simulation_statistics.json").read_text() == J["file_statistics_json"]
    assert (tmp_path / "ex" / "simulation_summary.json").read_text() == J["file_summary_json"]
    assert hashlib.sha256((tmp_path / "ex" / "frames" / "frame_0000.json").read_bytes()).hexdigest() == \
        J["file_frame_json_sha256"]
    assert sorted(p.name for p in (tmp_path / "ex" / "frames").iterdir()) == J["frames_dir"]
    assert J["export_bad_format"] == ["ValueError"] * 3
    for bad in (lambda: ex.export_frames(fr, "csv"), lambda: ex.export_statistics(sc.statistics, "csv"),
                lambda: ex.export_summary(sc, "txt")):
        with pytest.raises(ValueError):
            bad()
    (tmp_path / "s").mkdir()
    sc._save_simple_summary(tmp_path / "s")
    assert (tmp_path / "s" / "simulation_summary.txt").read_text(encoding="utf-8") == J["file_simple_summary_txt"]
    sc._export_combined_pointcloud_with_labels(tmp_path / "s")
    assert (tmp_path / "s" / "combined_pointcloud_with_label.ply").read_bytes() == A["file_labeled_ply"].tobytes()
    sc._save_labeled_ply(tmp_path / "x.ply", A["in_points_0"][:17], A["ply_in_colors"], A["ply_in_sem"], A["ply_in_ins"])
    assert (tmp_path / "x.ply").read_bytes() == A["file_x_ply"].tobytes()


def test_save_results_writes_the_reference_file_set(A, J, tmp_path):
    sc = S3DISSimScene("golden_room")
    for f in frames_from(A, J):
        sc.append_frame(f)
    sc.compute_statistics(2.5)
    sc.save_results(tmp_path / "out")                      # default formats: pkl (ignored, as in the reference) + txt
    names = sorted(p.name for p in (tmp_path / "out").iterdir())
    assert names == ["combined_pointcloud.ply", "combined_pointcloud_with_label.ply", "simulation_statistics.txt",
                     "simulation_summary.txt"]
    assert sc.statistics.simulation_time == 0.0            # recomputed inside save_results, reference quirk
    head = (tmp_path / "out" / "combined_pointcloud.ply").read_bytes()[:200]
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 125\nproperty double x\n")
    sc.save_results(tmp_path / "out2", formats=["json"])
    assert sorted(p.name for p in (tmp_path / "out2").iterdir()) == [
        "combined_pointcloud.ply", "combined_pointcloud_with_label.ply", "simulation_statistics.json",
        "simulation_summary.json"]

#!/usr/bin/env python3
"""Golden outputs of the reference's result containers (S3DISSimFrame / S3DISSimScene / ResultExporter), captured by
running the REFERENCE's own classes on seeded inputs (build container only).

    python tests/golden/make_containers_golden.py     # writes tests/golden/containers_golden.npz / .json

containers/s3dis_sim_frame.py and containers/s3dis_sim_scene.py are numpy-only; they are loaded under a synthetic
package so that the reference's containers/__init__.py (which pulls Open3D through s3dis_scene.py) is not executed.
The fixture holds inputs and outputs (arrays, returned dictionaries, the text / bytes of written files) only.
"""
import hashlib
import importlib.util
import json
import os
import sys
import tempfile
import types
from pathlib import Path

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402

pkg = types.ModuleType("refcontainers")
pkg.__path__ = [os.path.join(REF, "containers")]
sys.modules["refcontainers"] = pkg


def load(name):
    spec = importlib.util.spec_from_file_location(f"refcontainers.{name}", os.path.join(REF, "containers", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f"refcontainers.{name}"] = mod
    spec.loader.exec_module(mod)
    return mod


F = load("s3dis_sim_frame")
S = load("s3dis_sim_scene")


def make_inputs():
    rng = np.random.RandomState(42)
    frames = []
    for i, n in enumerate((40, 0, 25, 60)):
        pts = rng.uniform(-4, 6, size=(n, 3)).astype(np.float32)
        ang = rng.uniform(0, 90, size=n)
        q = dict(coverage_ratio=float(rng.uniform(0.2, 1.0)), num_points=n,
                 incident_angle_mean=float(ang.mean()) if n else 0.0, incident_angle_std=float(ang.std()) if n else 0.0,
                 scan_density=float(rng.uniform(10, 500)), range_mean=float(rng.uniform(1, 8)),
                 range_std=float(rng.uniform(0.1, 2)))
        frames.append((i * 3, pts, ang, q, {"waypoint": [float(i), 2.0, 1.0], "tag": f"f{i}"}))
    return frames


def jsonable(o):
    return json.loads(json.dumps(o, cls=S.NumpyEncoder))


def main():
    A, J = {}, {"numpy": np.__version__}
    inputs = make_inputs()
    ref_frames = []
    for k, (idx, pts, ang, q, md) in enumerate(inputs):
        A[f"in_points_{k}"], A[f"in_angles_{k}"] = pts, ang
        J[f"in_quality_{k}"], J[f"in_meta_{k}"], J[f"in_index_{k}"] = q, md, idx
        ref_frames.append(F.S3DISSimFrame(idx, pts, ang, F.ScanQuality(**q), dict(md)))

    # ---- frame level ----
    f0 = ref_frames[0]
    A["center_0"], A["std_0"] = f0.get_point_cloud_center(), f0.get_point_cloud_std()
    A["center_empty"], A["std_empty"] = ref_frames[1].get_point_cloud_center(), ref_frames[1].get_point_cloud_std()
    J["bounds_0"], J["bounds_empty"] = f0.get_point_cloud_bounds(), ref_frames[1].get_point_cloud_bounds()
    for name, g in (("angle", f0.filter_points_by_angle(20.0, 70.0)), ("angle_default", f0.filter_points_by_angle()),
                    ("range", f0.filter_points_by_range(2.0, 6.5)), ("range_none", f0.filter_points_by_range(100.0))):
        A[f"filt_{name}_points"], A[f"filt_{name}_angles"] = g.points, g.incident_angles
        J[f"filt_{name}_quality"] = jsonable(g.scan_quality.to_dict())
        J[f"filt_{name}_meta"] = g.frame_metadata
    try:
        ref_frames[1].filter_points_by_range(0.0, 1.0)
        J["filter_empty_frame"] = "ok"
    except Exception as e:                                        # noqa: BLE001
        J["filter_empty_frame"] = type(e).__name__
    d0 = f0.to_dict()
    J["frame0_dict_keys"] = sorted(d0)
    back = F.S3DISSimFrame.from_dict(json.loads(json.dumps(d0)))
    A["roundtrip_points_0"], A["roundtrip_angles_0"] = back.points, back.incident_angles
    J["roundtrip_quality_0"] = back.scan_quality.to_dict()
    J["repr_frame_0"] = repr(f0)
    ia = F.IncidentAngles(angles=inputs[0][2], surface_normals=inputs[0][1].astype(np.float64))
    hist, bins = ia.get_angle_distribution(7)
    A["ia_hist"], A["ia_bins"] = hist, bins
    J["ia_mean"], J["ia_std"] = float(ia.get_mean_angle()), float(ia.get_std_angle())
    ia2 = F.IncidentAngles.from_dict(json.loads(json.dumps(ia.to_dict())))
    J["ia_roundtrip_has"] = [ia2.surface_normals is not None, ia2.ray_directions is not None]

    # ---- scene level ----
    sc = S.S3DISSimScene("golden_room", {"lidar": "8line", "n": 4})
    J["empty_frame_statistics"], J["empty_quality_distribution"] = sc.get_frame_statistics(), sc.get_quality_distribution()
    sc.compute_statistics(3.0)
    J["empty_statistics"] = sc.statistics.to_dict()
    for f in ref_frames:
        sc.append_frame(f)
    J["frame_statistics"] = jsonable(sc.get_frame_statistics())
    J["quality_distribution"] = jsonable(sc.get_quality_distribution())
    sc.compute_statistics(2.5)
    J["statistics_2p5"] = jsonable(sc.statistics.to_dict())
    J["repr_scene"] = repr(sc)
    J["filter_frames_0p4_0p9"] = [f.frame_index for f in sc.filter_frames_by_quality(0.4, 0.9).frames]
    for metric in ("coverage", "points", "density"):
        J[f"best_{metric}"] = [f.frame_index for f in sc.get_best_frames(2, metric)]
    try:
        sc.get_best_frames(2, "nope")
    except Exception as e:                                        # noqa: BLE001
        J["best_bad_metric"] = type(e).__name__
    sd = sc.to_dict()
    J["scene_dict_keys"] = sorted(sd)
    sc2 = S.S3DISSimScene.from_dict(json.loads(json.dumps(sd, cls=S.NumpyEncoder)))
    J["scene_roundtrip"] = {"frames": sc2.get_total_frames(), "points": sc2.get_total_points(),
                            "statistics": jsonable(sc2.statistics.to_dict())}

    # ---- files ----
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        ex = S.ResultExporter(td / "ex")
        ex.export_statistics(sc.statistics, "txt")
        ex.export_statistics(sc.statistics, "json")
        ex.export_summary(sc, "json")
        ex.export_frames(ref_frames[:1], "json")
        ex.export_frames(ref_frames[2:3], "pkl")
        J["file_statistics_txt"] = (td / "ex" / "simulation_statistics.txt").read_text()
        J["file_statistics_json"] = (td / "ex" / "simulation_statistics.json").read_text()
        J["file_summary_json"] = (td / "ex" / "simulation_summary.json").read_text()
        J["file_frame_json_sha256"] = hashlib.sha256((td / "ex" / "frames" / "frame_0000.json").read_bytes()).hexdigest()
        J["frames_dir"] = sorted(p.name for p in (td / "ex" / "frames").iterdir())
        for bad in (lambda: ex.export_frames(ref_frames, "csv"), lambda: ex.export_statistics(sc.statistics, "csv"),
                    lambda: ex.export_summary(sc, "txt")):
            try:
                bad()
                J.setdefault("export_bad_format", []).append("ok")
            except Exception as e:                                # noqa: BLE001
                J.setdefault("export_bad_format", []).append(type(e).__name__)
        (td / "s").mkdir()
        sc._save_simple_summary(td / "s")
        J["file_simple_summary_txt"] = (td / "s" / "simulation_summary.txt").read_text(encoding="utf-8")
        sc._export_combined_pointcloud_with_labels(td / "s")
        ply = (td / "s" / "combined_pointcloud_with_label.ply").read_bytes()
        A["file_labeled_ply"] = np.frombuffer(ply, dtype=np.uint8).copy()
        rng = np.random.RandomState(7)
        n = 17
        cols = rng.randint(0, 256, size=(n, 3)).astype(np.uint8)
        sem, ins = rng.randint(0, 14, n).astype(np.uint16), rng.randint(0, 300, n).astype(np.uint16)
        A["ply_in_colors"], A["ply_in_sem"], A["ply_in_ins"] = cols, sem, ins
        sc._save_labeled_ply(td / "x.ply", inputs[0][1][:n], cols, sem, ins)
        A["file_x_ply"] = np.frombuffer((td / "x.ply").read_bytes(), dtype=np.uint8).copy()

    np.savez_compressed(os.path.join(OUT, "containers_golden.npz"), **A)
    with open(os.path.join(OUT, "containers_golden.json"), "w") as f:
        json.dump(J, f, indent=1, sort_keys=True)
    print("wrote", len(A), "arrays,", len(J), "json entries; filter on empty frame:", J["filter_empty_frame"])


if __name__ == "__main__":
    main()

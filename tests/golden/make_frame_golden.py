#!/usr/bin/env python3
"""Golden values of RobotPose / LidarPose / S3DISFrame from the REFERENCE's containers/s3dis_frame.py (numpy only,
loaded standalone; build container only).

    python tests/golden/make_frame_golden.py     # writes tests/golden/frame_golden.json
"""
import importlib.util
import json
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_s3dis_frame", os.path.join(REF, "containers", "s3dis_frame.py"))
F = importlib.util.module_from_spec(spec)
sys.modules["ref_s3dis_frame"] = F
spec.loader.exec_module(F)


def rot(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return rz @ ry @ rx


def main():
    G = {"cases": []}
    for k, (ypr, pos, mount_ypr, mount_pos) in enumerate((
            ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),
            ((0.7, 0.0, 0.0), (2.5, 2.0, 1.0), (0.0, 0.0, 0.0), (0.1, 0.0, 0.45)),
            ((-2.4, 0.3, -0.2), (-1.0, 4.0, 0.3), (1.57, -0.1, 0.05), (0.2, -0.1, 0.6)))):
        R, Rm = rot(*ypr), rot(*mount_ypr)
        robot = F.RobotPose(position=np.array(pos), orientation=R, timestamp=1.5 * k,
                            velocity=np.array([0.1 * k, 0.0, 0.0]) if k else None,
                            angular_velocity=np.array([0.0, 0.0, 0.2]) if k == 2 else None)
        fr = F.S3DISFrame(10 + k, robot, frame_metadata={"k": k})
        fr.add_lidar_pose("top", F.LidarPose(position=np.array(mount_pos), orientation=Rm, sensor_id="top"))
        d = fr.to_dict()
        back = F.S3DISFrame.from_dict(json.loads(json.dumps(d)))
        G["cases"].append({
            "ypr": ypr, "pos": pos, "mount_ypr": mount_ypr, "mount_pos": mount_pos,
            "robot_matrix": fr.get_robot_pose_matrix().tolist(),
            "yaw_pitch_roll": [float(robot.get_yaw()), float(robot.get_pitch()), float(robot.get_roll())],
            "global_default": fr.get_global_lidar_pose().tolist(),
            "global_top": fr.get_global_lidar_pose("top").tolist(),
            "sensors": fr.get_available_sensors(), "timestamp": fr.get_timestamp(), "repr": repr(fr),
            "dict": d,
            "roundtrip_dict": back.to_dict(),
            "roundtrip_twist_none": [back.robot_pose.velocity is None, back.robot_pose.angular_velocity is None],
            "from_matrix_dict": F.RobotPose.from_matrix(fr.get_robot_pose_matrix(), 3.0).to_dict(),
            "lidar_from_matrix_dict": F.LidarPose.from_matrix(fr.get_lidar_pose_matrix("top"), "x").to_dict(),
        })
    fr.remove_lidar_pose("top")
    fr.remove_lidar_pose("never_there")
    G["after_remove"] = fr.get_available_sensors()
    try:
        fr.get_lidar_position("top")
    except Exception as e:                                        # noqa: BLE001
        G["missing_sensor"] = type(e).__name__
    with open(os.path.join(OUT, "frame_golden.json"), "w") as f:
        json.dump(G, f, indent=1)
    print("wrote", len(G["cases"]), "cases")


if __name__ == "__main__":
    main()

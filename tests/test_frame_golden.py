"""RobotPose / LidarPose / S3DISFrame against values from the reference's own classes
(tests/golden/make_frame_golden.py), exact."""
import json
import os

import numpy as np
import pytest

from containers import LidarPose, RobotPose, S3DISFrame

HERE = os.path.dirname(os.path.abspath(__file__))


def rot(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return rz @ ry @ rx


def test_frame_records_match_the_reference():
    with open(os.path.join(HERE, "golden", "frame_golden.json")) as f:
        G = json.load(f)
    for k, c in enumerate(G["cases"]):
        robot = RobotPose(position=np.array(c["pos"]), orientation=rot(*c["ypr"]), timestamp=1.5 * k,
                          velocity=np.array([0.1 * k, 0.0, 0.0]) if k else None,
                          angular_velocity=np.array([0.0, 0.0, 0.2]) if k == 2 else None)
        fr = S3DISFrame(10 + k, robot, frame_metadata={"k": k})
        fr.add_lidar_pose("top", LidarPose(position=np.array(c["mount_pos"]), orientation=rot(*c["mount_ypr"]),
                                           sensor_id="top"))
        assert fr.get_robot_pose_matrix().tolist() == c["robot_matrix"]
        assert [float(robot.get_yaw()), float(robot.get_pitch()), float(robot.get_roll())] == c["yaw_pitch_roll"]
        assert fr.get_global_lidar_pose().tolist() == c["global_default"]
        assert fr.get_global_lidar_pose("top").tolist() == c["global_top"]
        assert fr.get_available_sensors() == c["sensors"] and fr.get_timestamp() == c["timestamp"]
        assert repr(fr) == c["repr"]
        d = fr.to_dict()
        assert d == c["dict"]
        back = S3DISFrame.from_dict(json.loads(json.dumps(d)))
        assert back.to_dict() == c["roundtrip_dict"]
        assert [back.robot_pose.velocity is None, back.robot_pose.angular_velocity is None] == c["roundtrip_twist_none"]
        assert RobotPose.from_matrix(fr.get_robot_pose_matrix(), 3.0).to_dict() == c["from_matrix_dict"]
        assert LidarPose.from_matrix(fr.get_lidar_pose_matrix("top"), "x").to_dict() == c["lidar_from_matrix_dict"]
        assert np.array_equal(fr.get_robot_position(), robot.position)
        assert np.array_equal(fr.get_lidar_orientation("top"), rot(*c["mount_ypr"]))
    fr.remove_lidar_pose("top")
    fr.remove_lidar_pose("never_there")
    assert fr.get_available_sensors() == G["after_remove"]
    assert G["missing_sensor"] == "ValueError"
    for call in (fr.get_lidar_position, fr.get_lidar_orientation, fr.get_lidar_pose_matrix, fr.get_global_lidar_pose):
        with pytest.raises(ValueError):
            call("top")

"""Input-side frame record: where the robot is and how its sensors sit on it.

Mirrors the public surface of the reference's ``containers/s3dis_frame.py`` (RobotPose, LidarPose, S3DISFrame); the
values every method returns are checked against the reference's own classes (tests/golden/make_frame_golden.py,
tests/test_frame_golden.py).  How this feeds the scan: ``S3DISFrame.get_global_lidar_pose()`` -- robot pose times
mounting pose -- is the (4,4) matrix ``lidar.create_lidar`` / ``RaycastEngineGPU.scan_poses`` take per waypoint.
"""
from dataclasses import dataclass
from typing import Any, Dict, Optional

import numpy as np


@dataclass
class _RigidTransform:
    """position (3,) + rotation matrix (3,3); the two pose records below add their own extra fields."""
    position: np.ndarray
    orientation: np.ndarray

    def to_matrix(self) -> np.ndarray:
        """Homogeneous (4,4) form: rotation in the upper-left block, translation in the last column."""
        out = np.eye(4)
        out[:3, 3] = self.position
        out[:3, :3] = self.orientation
        return out

    def _base_dict(self) -> Dict[str, Any]:
        return {"position": self.position.tolist(), "orientation": self.orientation.tolist()}


@dataclass
class LidarPose(_RigidTransform):
    """Mounting pose of one sensor, expressed in the robot frame."""
    sensor_id: str = "lidar_0"

    @classmethod
    def from_matrix(cls, matrix: np.ndarray, sensor_id: str = "lidar_0") -> "LidarPose":
        return cls(position=matrix[:3, 3], orientation=matrix[:3, :3], sensor_id=sensor_id)

    def to_dict(self) -> Dict[str, Any]:
        return {**self._base_dict(), "sensor_id": self.sensor_id}


@dataclass
class RobotPose(_RigidTransform):
    """World pose of the robot base with its time stamp and, optionally, linear / angular velocity (3,)."""
    timestamp: float = 0.0
    velocity: Optional[np.ndarray] = None
    angular_velocity: Optional[np.ndarray] = None

    @classmethod
    def from_matrix(cls, matrix: np.ndarray, timestamp: float = 0.0) -> "RobotPose":
        return cls(position=matrix[:3, 3], orientation=matrix[:3, :3], timestamp=timestamp)

    # Z-Y-X Euler angles read off the rotation matrix R = Rz(yaw) Ry(pitch) Rx(roll)
    def get_roll(self) -> float:
        return np.arctan2(self.orientation[2, 1], self.orientation[2, 2])

    def get_pitch(self) -> float:
        r20, r21, r22 = self.orientation[2]
        return np.arctan2(-r20, np.sqrt(r21 ** 2 + r22 ** 2))

    def get_yaw(self) -> float:
        return np.arctan2(self.orientation[1, 0], self.orientation[0, 0])

    def to_dict(self) -> Dict[str, Any]:
        twist = {name: (None if val is None else val.tolist())
                 for name, val in (("velocity", self.velocity), ("angular_velocity", self.angular_velocity))}
        return {**self._base_dict(), "timestamp": self.timestamp, **twist}


class S3DISFrame:
    """One time step of the platform: frame index, robot pose, sensors by id, free-form metadata.

    Without an explicit sensor table the frame carries a single sensor "lidar_0" mounted at the robot origin."""

    def __init__(self, frame_index: int, robot_pose: RobotPose, lidar_poses: Optional[Dict[str, LidarPose]] = None,
                 frame_metadata: Optional[Dict[str, Any]] = None):
        self.frame_index = frame_index
        self.robot_pose = robot_pose
        if not lidar_poses:
            lidar_poses = {"lidar_0": LidarPose(position=np.array([0, 0, 0]), orientation=np.eye(3))}
        self.lidar_poses = lidar_poses
        self.frame_metadata = frame_metadata or {}

    # ---- sensor table -------------------------------------------------------------------------------------
    def get_available_sensors(self) -> list:
        return list(self.lidar_poses)

    def add_lidar_pose(self, sensor_id: str, lidar_pose: LidarPose):
        self.lidar_poses[sensor_id] = lidar_pose

    def remove_lidar_pose(self, sensor_id: str):
        """Forget a sensor; unknown ids are ignored."""
        self.lidar_poses.pop(sensor_id, None)

    def _mounted(self, sensor_id: str) -> LidarPose:
        try:
            return self.lidar_poses[sensor_id]
        except KeyError:
            raise ValueError(f"LiDAR sensor {sensor_id} does not exist") from None

    def get_lidar_position(self, sensor_id: str = "lidar_0") -> np.ndarray:
        return self._mounted(sensor_id).position

    def get_lidar_orientation(self, sensor_id: str = "lidar_0") -> np.ndarray:
        return self._mounted(sensor_id).orientation

    def get_lidar_pose_matrix(self, sensor_id: str = "lidar_0") -> np.ndarray:
        return self._mounted(sensor_id).to_matrix()

    # ---- robot --------------------------------------------------------------------------------------------
    def get_timestamp(self) -> float:
        return self.robot_pose.timestamp

    def get_robot_position(self) -> np.ndarray:
        return self.robot_pose.position

    def get_robot_orientation(self) -> np.ndarray:
        return self.robot_pose.orientation

    def get_robot_pose_matrix(self) -> np.ndarray:
        return self.robot_pose.to_matrix()

    def get_global_lidar_pose(self, sensor_id: str = "lidar_0") -> np.ndarray:
        """World pose of a sensor = robot pose @ mounting pose: the matrix the ray generators take."""
        return self.get_robot_pose_matrix() @ self.get_lidar_pose_matrix(sensor_id)

    # ---- dictionary round trip ----------------------------------------------------------------------------
    def to_dict(self) -> Dict[str, Any]:
        return {"frame_index": self.frame_index, "robot_pose": self.robot_pose.to_dict(),
                "lidar_poses": {sid: pose.to_dict() for sid, pose in self.lidar_poses.items()},
                "frame_metadata": self.frame_metadata}

    @classmethod
    def from_dict(cls, frame_dict: Dict[str, Any]) -> "S3DISFrame":
        src = frame_dict["robot_pose"]
        twist = {key: (np.array(src[key]) if src[key] else None)          # absent or empty -> None
                 for key in ("velocity", "angular_velocity")}
        robot = RobotPose(position=np.array(src["position"]), orientation=np.array(src["orientation"]),
                          timestamp=src["timestamp"], **twist)
        mounted = {sid: LidarPose(position=np.array(rec["position"]), orientation=np.array(rec["orientation"]),
                                  sensor_id=sid)
                   for sid, rec in frame_dict["lidar_poses"].items()}
        return cls(frame_index=frame_dict["frame_index"], robot_pose=robot, lidar_poses=mounted,
                   frame_metadata=frame_dict.get("frame_metadata", {}))

    def __repr__(self) -> str:
        return (f"S3DISFrame(index={self.frame_index}, timestamp={self.get_timestamp():.3f}, "
                f"sensors={self.get_available_sensors()})")

"""Records of the scan path, under the names the reference's ``containers`` package exports.

    scene           S3DISScene (mesh + RoomBounds + SemanticInfo)              s3dis_scene.py
    input frame     S3DISFrame (RobotPose + LidarPose per sensor)             s3dis_frame.py
    scan output     S3DISSimFrame (points, incident angles, ScanQuality,      s3dis_sim_frame.py
                    optional per-point labels), IncidentAngles
    scene output    S3DISSimScene (frames -> statistics -> result files),      s3dis_sim_scene.py
                    SimulationStats, ResultExporter, NumpyEncoder, labelled-PLY helpers
"""
from .s3dis_sim_scene import (NumpyEncoder, ResultExporter, S3DISSimScene, SimulationStats, read_labeled_ply,
                              write_labeled_ply)
from .s3dis_sim_frame import IncidentAngles, S3DISSimFrame, ScanQuality
from .s3dis_frame import LidarPose, RobotPose, S3DISFrame
from .s3dis_scene import RoomBounds, S3DISScene, SemanticInfo

__all__ = ["IncidentAngles", "LidarPose", "NumpyEncoder", "ResultExporter", "RobotPose", "RoomBounds", "S3DISFrame",
           "S3DISScene", "S3DISSimFrame", "S3DISSimScene", "ScanQuality", "SemanticInfo", "SimulationStats",
           "read_labeled_ply", "write_labeled_ply"]

"""Containers the scan output is poured into (subset of the reference's ``containers`` package)."""
from .s3dis_scene import S3DISScene, RoomBounds, SemanticInfo
from .s3dis_sim_frame import S3DISSimFrame, ScanQuality, IncidentAngles
from .s3dis_sim_scene import (S3DISSimScene, SimulationStats, ResultExporter, NumpyEncoder, write_labeled_ply,
                              read_labeled_ply)

__all__ = ["S3DISScene", "RoomBounds", "SemanticInfo", "S3DISSimFrame", "ScanQuality", "IncidentAngles", "S3DISSimScene",
           "SimulationStats", "ResultExporter", "NumpyEncoder", "write_labeled_ply", "read_labeled_ply"]

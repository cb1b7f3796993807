#!/usr/bin/env python3
"""The engine boundary alone: the two calls of the reference's RaycastEngineBase on the HIP engine.

    python examples/drop_in_engine.py
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "indoor-point-cloud-datasets-controllable-generation-method-for-mobile-"
                                      "robots-3d-scene-perception_amd"))

import numpy as np  # noqa: E402
from lidar import Indoor8LineLidarIntrinsics, create_lidar  # noqa: E402
from lidarcast import synth  # noqa: E402
from raycast_engine import RaycastEngineGPU  # noqa: E402
from trajectory import Waypoint  # noqa: E402

mesh = synth.make_room(size=(5.0, 4.0, 2.8), num_boxes=4, seed=1, cell=0.05)     # .vertices / .triangles, like Open3D's
engine = RaycastEngineGPU()                                                       # raises without an MI355X
lidar = create_lidar(Indoor8LineLidarIntrinsics.create_standard_8line(), Waypoint(2.5, 2.0, 1.0, 0.3).to_pose_matrix())

points, incident_angles = engine.lidar_intersect_mesh(lidar, mesh)                # (K,3) float32, (K,) float64
print("lidar_intersect_mesh:", points.shape, points.dtype, incident_angles.shape, incident_angles.dtype)

rays = lidar.get_rays()[:1000]                                                    # any (N,6) array of origins|directions
hit_points = engine.rays_intersect_mesh(rays, mesh)
print("rays_intersect_mesh :", hit_points.shape, "of", len(rays), "rays hit")

try:
    engine.rays_intersect_mesh(rays[:, :5], mesh)
except ValueError as e:
    print("bad input ->", type(e).__name__, e)

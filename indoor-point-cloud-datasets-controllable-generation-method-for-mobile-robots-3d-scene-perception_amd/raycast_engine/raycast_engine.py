"""Engine plugin boundary (reference: raycast_engine/raycast_engine.py:16-61).

Same abstract interface as the reference; the difference is stated in the class note: engines
here keep the scene (BVH) of a mesh between calls instead of rebuilding it for every pose.
"""
from abc import ABC, abstractmethod

import numpy as np


class RaycastEngineBase(ABC):
    """Abstract ray-cast engine: mesh-ray intersection for explicit rays and for a LiDAR pose.

    Notes:
        - The reference assumes "a scene is only used for raycasting once" and rebuilds it per call.
          Engines of this package cache the scene per mesh; ``clear_cache()`` drops it.
    """

    @abstractmethod
    def __init__(self):
        pass

    @abstractmethod
    def rays_intersect_mesh(self, rays: np.ndarray, mesh):
        """rays (N, 6) float32, mesh with ``.vertices`` / ``.triangles`` -> hit points (K, 3) float32."""

    @abstractmethod
    def lidar_intersect_mesh(self, lidar, mesh):
        """lidar (``get_rays()``, ``pose``, ``intrinsics.max_range``), mesh ->
        (points (K, 3) float32, incident_angles (K,) float64)."""

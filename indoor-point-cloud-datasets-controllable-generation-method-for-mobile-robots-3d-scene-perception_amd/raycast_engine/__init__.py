"""Ray-cast engines (import surface of the reference's ``raycast_engine`` package)."""
from .raycast_engine import RaycastEngineBase
from .raycast_engine_hip import RaycastEngineHIP, RaycastEngineCPU, RaycastEngineGPU, mesh_arrays

__all__ = ["RaycastEngineBase", "RaycastEngineCPU", "RaycastEngineGPU", "RaycastEngineHIP",
           "mesh_arrays"]

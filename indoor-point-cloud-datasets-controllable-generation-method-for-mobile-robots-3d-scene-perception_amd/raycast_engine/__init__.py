"""Ray-cast engines behind the reference's engine interface.  ``RaycastEngineGPU`` and ``RaycastEngineCPU`` are both
the MI355X engine (there is no CPU compute path in this package); ``RaycastEngineBase`` is the abstract interface."""
from . import raycast_engine as _base
from . import raycast_engine_hip as _hip

RaycastEngineBase = _base.RaycastEngineBase
RaycastEngineHIP, RaycastEngineGPU, RaycastEngineCPU = _hip.RaycastEngineHIP, _hip.RaycastEngineGPU, _hip.RaycastEngineCPU
mesh_arrays = _hip.mesh_arrays

__all__ = ["RaycastEngineBase", "RaycastEngineCPU", "RaycastEngineGPU", "RaycastEngineHIP", "mesh_arrays"]

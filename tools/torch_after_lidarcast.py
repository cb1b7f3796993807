#!/usr/bin/env python3
"""tools/torch_after_lidarcast.py -- this library's HIP context first, torch's afterwards, in one process (the order the
parity tests take when a single test is selected): both must see the GPU (lidarcast/_capi.py::_share_torch_hip_runtime)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "indoor-point-cloud-datasets-controllable-generation-method-for-mobile-robots-3d-scene-perception_amd"))
import lidarcast  # noqa: E402

ctx = lidarcast.Context(0)
print("lidarcast context ok; torch imported:", "torch" in sys.modules)
import torch  # noqa: E402

x = torch.zeros(4, device="cuda")
torch.cuda.synchronize()
print("torch after lidarcast ok:", x.sum().item(), torch.cuda.get_device_name(0))
for line in open("/proc/self/maps"):
    if "libamdhip64" in line and "r-xp" in line:
        print("runtime mapped:", line.split()[-1])

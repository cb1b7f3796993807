import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from helpers import pose, sensor_small
from raycast_engine import RaycastEngineGPU
from lidarcast import synth
e = RaycastEngineGPU()
mesh = synth.make_room(size=(4, 3, 2.5), num_boxes=4, seed=5, cell=0.05)
for k in (sensor_small(lines=4, width=64, max_range=1.05), sensor_small(lines=5, width=200, max_range=1.9)):
    poses = np.stack([pose(0.6 + 0.45 * i, 1.2 + 0.1 * i, 1.0, 0.37 * i) for i in range(7)])
    fr = e.scan_frames(k, poses, mesh, want=("point3", "incident_deg", "range_origin", "range_origin_stats", "incident_stats"))
    rng_f, ang_f = e.split_frames(fr, "range_origin"), e.split_frames(fr, "incident_deg")
    for i in range(7):
        n = int(fr["counts"][i])
        if n == 0: continue
        print(n, "range mean", fr["range_origin_mean"][i], np.mean(rng_f[i]), "std", fr["range_origin_std"][i], np.std(rng_f[i]),
              "| inc mean %.17g %.17g" % (fr["incident_mean"][i], np.mean(ang_f[i])), "std %.17g %.17g" % (fr["incident_std"][i], np.std(ang_f[i])))

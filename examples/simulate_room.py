#!/usr/bin/env python3
"""End-to-end use of the drop-in surface on an MI355X: room mesh -> planned trajectory -> pose-batched scan ->
result files, with the reference's class and method names.

    python examples/simulate_room.py [--mesh room.ply] [--out output_dir] [--waypoints 20]

Without a mesh path a procedural room (2 cm tessellation, labelled triangles) stands in for a reconstructed S3DIS room.
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "indoor-point-cloud-datasets-controllable-generation-method-for-mobile-"
                                      "robots-3d-scene-perception_amd"))

import numpy as np  # noqa: E402
from s3dis_simulator import S3DISSimulator  # noqa: E402  (same module name as the reference's)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default=None, help="triangle-mesh PLY of a room (default: a procedural room)")
    ap.add_argument("--out", default="simulation_results_example")
    ap.add_argument("--waypoints", type=int, default=20)
    args = ap.parse_args()
    mesh_path, out_dir = args.mesh, args.out
    sim = S3DISSimulator({"raycast_engine": {"use_gpu": True}, "trajectory": {"robot_height": 1.0}},
                         use_dense_lidar=True)                     # 32 lines x 4000 azimuths, 25 m
    if mesh_path is None:
        from lidarcast import synth
        scene = sim.load_scene(synth.make_room(size=(6.0, 4.5, 2.8), num_boxes=6, seed=3, cell=0.02), "procedural_room")
    else:
        scene = sim.load_scene(mesh_path)
    print(f"scene {scene.scene_name}: {scene.num_vertices} vertices, {scene.num_triangles} triangles")

    np.random.seed(0)                                              # the planner draws its candidates from np.random
    t0 = time.perf_counter()
    waypoints, analysis = sim.generate_auto_trajectory(num_waypoints=args.waypoints)
    print(f"planned {len(waypoints)} waypoints in {time.perf_counter() - t0:.2f} s "
          f"({analysis.get('total_candidates', '?')} candidates)")

    t0 = time.perf_counter()
    sim_scene = sim.run_simulation(waypoints)                      # one launch for the whole trajectory
    dt = time.perf_counter() - t0
    rays = len(waypoints) * sim.lidar_config.get_total_points_per_scan()
    print(f"scanned {rays} rays -> {sim_scene.get_total_points()} points in {dt * 1e3:.1f} ms "
          f"({rays / dt / 1e6:.0f} M rays/s incl. scene build and host copies)")

    sim.save_results(sim_scene, out_dir, waypoints)
    print("wrote", sorted(os.listdir(out_dir)))


if __name__ == "__main__":
    main()

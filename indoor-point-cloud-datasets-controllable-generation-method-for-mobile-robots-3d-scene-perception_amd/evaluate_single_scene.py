"""Distribution-similarity metrics between two point clouds, under the reference's module and function names
(reference: evaluate_single_scene.py:15-209).  The O(n*m) parts run as HIP kernels (csrc/lrc_metrics.hip through
lidarcast.metrics); sampling and definitions are the reference's, checked against values computed with the
reference's own functions (tests/golden/make_metrics_golden.py).  The LiDAR-Net scene-matching CLI around them
(find_lidar_net_scenes, find_best_match, main) is out of scope.
"""
from lidarcast.metrics import (analyze_point_cloud, check_volume_compatibility, compute_chamfer_distance,  # noqa: F401
                               compute_hausdorff_distance, compute_mmd_sampled, evaluate_clouds,
                               normalize_coordinates, sample_points)


def load_point_cloud(ply_path):
    """(N,3) coordinates of a PLY point cloud, None if it cannot be read (the reference prints and returns None)."""
    try:
        from lidarcast.ply import read_point_cloud
        return read_point_cloud(ply_path)
    except Exception as e:                                         # noqa: BLE001
        print(f"[Error] Failed to load point cloud {ply_path}: {e}")
        return None


def evaluate_single_scene(s3dis_ply, lidar_net_ply, max_points=10000, volume_threshold=0.3):
    """MMD / Chamfer / Hausdorff / density ratio of two PLY clouds after centring each on its bounding box; None when
    a file cannot be read or the bounding-box volumes differ by more than the threshold (reference :165-209)."""
    a, b = load_point_cloud(s3dis_ply), load_point_cloud(lidar_net_ply)
    if a is None or b is None:
        return None
    return evaluate_clouds(a, b, max_points=max_points, volume_threshold=volume_threshold)

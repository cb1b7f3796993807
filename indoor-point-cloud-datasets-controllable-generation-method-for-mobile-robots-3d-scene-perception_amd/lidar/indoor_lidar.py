"""Ray generators of the two sensor models, vectorised.

Same outputs as the reference generators (float32 (N,6) rows [origin | direction], line-major and
azimuth-minor), checked bit for bit against vectors captured from the reference
(tests/golden/, tests/test_lidar_golden.py):
  * IndoorLidar                     reference lidar/indoor_lidar.py:11-131
  * DualAxisLidar.get_multi_line_rays        reference lidar/indoor_lidar.py:224-296
    -- the reference draws two scalar normals per ray inside a 64 000-iteration Python loop and one
    uniform vector afterwards; one normal(size=2N) + one random(N) call consumes the legacy
    MT19937 stream identically, which is what is done here.
``sensor_directions()`` is the pose-independent float64 direction table the HIP scan kernel
rotates per pose (lrc_scan_poses).
"""
from dataclasses import dataclass

import numpy as np

from .lidar_intrinsics import DualAxisLidarIntrinsics, Indoor8LineLidarIntrinsics


def _check_pose(pose):
    assert isinstance(pose, np.ndarray)
    assert pose.shape == (4, 4)


@dataclass
class IndoorLidar:
    intrinsics: Indoor8LineLidarIntrinsics
    pose: np.ndarray

    def __post_init__(self):
        assert isinstance(self.intrinsics, Indoor8LineLidarIntrinsics)
        _check_pose(self.pose)

    # -- direction tables (sensor frame) ----------------------------------------------------------
    @staticmethod
    def directions_from_vertical_degrees(vertical_degrees, W):
        """(H*W,3) float64; azimuth beta = -(i - W/2)/W*2pi so column 0 looks along -x."""
        W = max(1, int(W))
        degs = list(vertical_degrees) if vertical_degrees is not None else []
        if len(degs) == 0:
            degs = [0.0]
        H = len(degs)
        beta = -(np.arange(W) - W / 2) / W * 2 * np.pi
        alpha = np.array([np.deg2rad(v) for v in degs])
        ca, sa = np.cos(alpha), np.sin(alpha)
        out = np.empty((H, W, 3), dtype=np.float64)
        out[..., 0] = ca[:, None] * np.cos(beta)[None, :]
        out[..., 1] = ca[:, None] * np.sin(beta)[None, :]
        out[..., 2] = sa[:, None]
        return out.reshape(H * W, 3)

    @staticmethod
    def directions_uniform(fov_up, fov_down, H, W):
        """(H*W,3) FLOAT32 table of the uniform-elevation branch (the reference narrows before rotating)."""
        H, W = max(1, int(H)), max(1, int(W))
        v = np.linspace(np.deg2rad(fov_up), -np.deg2rad(fov_down), H)
        h = np.linspace(0, 2 * np.pi, W, endpoint=False)
        out = np.empty((H, W, 3), dtype=np.float64)
        out[..., 0] = np.cos(v)[:, None] * np.cos(h)[None, :]
        out[..., 1] = np.cos(v)[:, None] * np.sin(h)[None, :]
        out[..., 2] = np.sin(v)[:, None]
        return out.reshape(H * W, 3).astype(np.float32)

    def sensor_directions(self):
        """float64 (N,3) pose-independent direction table for the in-kernel generator."""
        k = self.intrinsics
        if k.vertical_degrees is None:      # the reference narrows this branch's table to float32 before rotating
            return self.directions_uniform(k.fov_up, k.fov_down, k.vertical_res, k.horizontal_res).astype(np.float64)
        return self.directions_from_vertical_degrees(k.vertical_degrees, k.horizontal_res)

    # -- world-frame rays ---------------------------------------------------------------------------
    def get_rays(self) -> np.ndarray:
        k = self.intrinsics
        R, c = self.pose[:3, :3], self.pose[:3, 3]
        if k.vertical_degrees is None:
            d32 = self.directions_uniform(k.fov_up, k.fov_down, k.vertical_res, k.horizontal_res)
            world = (R @ d32.T).T.astype(np.float32)
            origins = np.tile(c, (len(d32), 1)).astype(np.float32)
        else:
            d64 = self.directions_from_vertical_degrees(k.vertical_degrees, k.horizontal_res)
            world = np.dot(d64, R.T).astype(np.float32)
            origins = np.expand_dims(c, axis=0).repeat(len(d64), axis=0).astype(np.float32)
        return np.concatenate([origins, world], axis=-1)

    # the reference's generator entry points, same names and (origins, directions) float32 return
    @staticmethod
    def _gen_lidar_rays(pose, fov_up, fov_down, H, W):
        """Uniform-elevation branch (lidar/indoor_lidar.py:56-91)."""
        d32 = IndoorLidar.directions_uniform(fov_up, fov_down, H, W)
        world = (pose[:3, :3] @ d32.T).T.astype(np.float32)
        return np.tile(pose[:3, 3], (len(d32), 1)).astype(np.float32), world

    @staticmethod
    def _gen_lidar_rays_with_vertical_degrees(pose, vertical_degrees, W):
        """Listed-elevation branch (lidar/indoor_lidar.py:94-131)."""
        d64 = IndoorLidar.directions_from_vertical_degrees(vertical_degrees, W)
        world = np.dot(d64, pose[:3, :3].T).astype(np.float32)
        return np.expand_dims(pose[:3, 3], axis=0).repeat(len(d64), axis=0).astype(np.float32), world

    def get_total_rays(self) -> int:
        k = self.intrinsics
        if k.vertical_degrees is None:
            return max(1, int(k.vertical_res)) * max(1, int(k.horizontal_res))
        return max(1, len(k.vertical_degrees)) * max(1, int(k.horizontal_res))

    def get_scan_frequency(self) -> float:
        return self.intrinsics.get_scan_frequency()

    def get_range_limits(self) -> tuple:
        return self.intrinsics.get_range_limits()


@dataclass
class DualAxisLidar:
    intrinsics: DualAxisLidarIntrinsics
    pose: np.ndarray
    rng: object = None   # None = the global numpy stream, as in the reference

    def __post_init__(self):
        assert isinstance(self.intrinsics, DualAxisLidarIntrinsics)
        _check_pose(self.pose)

    def scan_pattern(self, num_points=None):
        """Noise-free (phi, theta) of the scan, float64 (N,) each: a function of the intrinsics alone, computed once per
        intrinsics object and scan size (callers must not modify the arrays)."""
        k = self.intrinsics
        if num_points is None:
            num_points = int(k.point_rate * k.scan_duration)
        key = (num_points, k.num_vertical_lines, tuple(k.theta_range), k.swing_amplitude, k.swing_frequency)
        cached = getattr(k, "_scan_pattern", None)
        if cached is None or cached[0] != key:
            L = k.num_vertical_lines
            per_line = num_points // L
            base = np.linspace(k.theta_range[1], k.theta_range[0], L)
            phi = np.linspace(0, 2 * np.pi, per_line, endpoint=False)
            phase = np.arange(L) * np.pi / L
            theta = base[:, None] + k.swing_amplitude * np.sin(k.swing_frequency * phi[None, :] + phase[:, None])
            theta = np.clip(theta, k.theta_range[0], k.theta_range[1]).reshape(-1)
            phi = np.broadcast_to(phi[None, :], (L, per_line)).reshape(-1).copy()
            cached = (key, phi, theta)
            try:
                k._scan_pattern = cached
            except Exception:          # frozen / slotted intrinsics: no cache
                pass
        return cached[1], cached[2]

    def scan_angles(self, num_points=None):
        """Noisy (phi, theta) per ray, float64 (N,), and the dropout keep mask.  Consumes the RNG."""
        k = self.intrinsics
        rnd = np.random if self.rng is None else self.rng
        phi, theta = self.scan_pattern(num_points)
        phi, theta = phi.copy(), theta.copy()
        n = phi.size
        if k.angle_noise_std > 0:
            z = rnd.normal(0, k.angle_noise_std, size=2 * n).reshape(n, 2)
            phi += z[:, 0]
            theta += z[:, 1]
        keep = None
        if k.dropout_probability > 0:
            keep = rnd.random(n) > k.dropout_probability
        return phi, theta, keep

    def scan_angles_from_draws(self, z, u, num_points=None):
        """The same from draws made elsewhere (lidarcast.nprandom.scan_draws: a whole trajectory's draws of the seeded
        stream in one native call): z = the 2N normals of this pose (or None), u = its N uniforms (or None)."""
        k = self.intrinsics
        phi, theta = self.scan_pattern(num_points)
        phi, theta = phi.copy(), theta.copy()
        if z is not None:
            z = z.reshape(phi.size, 2)
            phi += z[:, 0]
            theta += z[:, 1]
        keep = None if u is None else u > k.dropout_probability
        return phi, theta, keep

    def all_rays_and_mask(self, num_points: int = None, out: np.ndarray = None):
        """Every ray of the scan BEFORE the dropout, (N,6) float32 (written into ``out`` when given, e.g. a page-locked
        buffer), and the dropout keep mask (N,) bool or None.  ``rays[keep]`` is ``get_multi_line_rays()``; consumes the
        RNG exactly as it does.  The batched engine path casts all rays with the mask instead of a ragged subset."""
        phi, theta, keep = self.scan_angles(num_points)
        return self.rays_from_angles(phi, theta, out), keep

    def rays_from_angles(self, phi, theta, out: np.ndarray = None):
        """(N,6) float32 rays of the scan angles (phi, theta): the trigonometry, rotation and narrowing of the reference's
        generator (lidar/indoor_lidar.py:274-291), no random draws -- callers that scan many poses draw the angles pose
        after pose (the seeded stream is sequential by definition) and run this part on a thread pool."""
        ct = np.cos(theta)
        d = np.stack([ct * np.cos(phi), ct * np.sin(phi), np.sin(theta)], axis=-1)
        R = self.pose[:3, :3]
        # the reference rotates ray by ray (R @ d); row form of the same product
        world = (d[:, 0:1] * R[:, 0][None, :] + d[:, 1:2] * R[:, 1][None, :]) + d[:, 2:3] * R[:, 2][None, :]
        rays = np.empty((len(d), 6), dtype=np.float32) if out is None else out
        rays[:, :3] = self.pose[:3, 3].astype(np.float32)
        rays[:, 3:] = world.astype(np.float32)
        return rays

    def get_multi_line_rays(self, num_points: int = None) -> np.ndarray:
        rays, keep = self.all_rays_and_mask(num_points)
        if keep is not None:
            rays = rays[keep]
        return rays

    def get_rays(self) -> np.ndarray:
        return self.get_multi_line_rays()

    # -- single-line time-sampled generators (reference lidar/indoor_lidar.py:162-222, :298-340) -------------
    def _rotate(self, d):
        """Rows of d (n,3) float64 through the pose rotation, as the reference's per-ray ``R @ d``."""
        R = self.pose[:3, :3]
        return (d[:, 0:1] * R[:, 0][None, :] + d[:, 1:2] * R[:, 1][None, :]) + d[:, 2:3] * R[:, 2][None, :]

    def get_rays_sequence(self, time_sequence: np.ndarray) -> np.ndarray:
        """(N,6) float32 rays of scan line 0 at the given times.  The reference loops over the times and draws
        two scalar normals per step (phi's, then theta's); one normal(size=2N) call consumes the stream alike."""
        k = self.intrinsics
        rnd = np.random if self.rng is None else self.rng
        t = np.asarray(time_sequence, dtype=np.float64).reshape(-1)
        phi, theta = k.swing_angles(t, 0)
        if k.angle_noise_std > 0 and t.size:
            z = rnd.normal(0, k.angle_noise_std, size=2 * t.size).reshape(t.size, 2)
            phi, theta = phi + z[:, 0], theta + z[:, 1]
        ct = np.cos(theta)
        d = np.stack([ct * np.cos(phi), ct * np.sin(phi), np.sin(theta)], axis=-1)
        rays = np.empty((t.size, 6), dtype=np.float32)
        rays[:, :3] = self.pose[:3, 3].astype(np.float32)
        rays[:, 3:] = self._rotate(d).astype(np.float32)
        return rays

    def get_rays_at_time(self, t: float) -> np.ndarray:
        """(1,6) float32: the ray of scan line 0 at time t.  Unlike the sequence form the reference narrows the
        sensor-frame direction to float32 BEFORE rotating it."""
        phi, theta = self.intrinsics.calculate_angles_at_time(t, line_idx=0)
        d32 = np.array([np.cos(theta) * np.cos(phi), np.cos(theta) * np.sin(phi), np.sin(theta)], dtype=np.float32)
        world = self._rotate(d32.astype(np.float64)[None, :])[0].astype(np.float32)
        return np.concatenate([self.pose[:3, 3].astype(np.float32), world]).reshape(1, 6)

    def get_rays_frame(self, frame_duration: float = None) -> np.ndarray:
        return self.get_rays_sequence(self.intrinsics.generate_time_sequence(frame_duration))

    def get_spiral_scan_rays(self, num_points: int = None):
        """(rays, timestamps): num_points samples spread over one scan_duration, end point included."""
        k = self.intrinsics
        if num_points is None:
            num_points = int(k.point_rate * k.scan_duration)
        stamps = np.linspace(0, k.scan_duration, num_points)
        return self.get_rays_sequence(stamps), stamps

    def add_noise_to_rays(self, rays: np.ndarray) -> np.ndarray:
        """Dropout only: one uniform draw per ray from the global stream (lidar/indoor_lidar.py:354-370)."""
        if self.intrinsics.dropout_probability > 0:
            rnd = np.random if self.rng is None else self.rng
            rays = rays[rnd.random(len(rays)) > self.intrinsics.dropout_probability]
        return rays

    def get_total_rays(self) -> int:
        return int(self.intrinsics.point_rate * self.intrinsics.scan_duration)

    def get_scan_frequency(self) -> float:
        return 1.0 / self.intrinsics.scan_duration

    def get_range_limits(self) -> tuple:
        return (0.5, self.intrinsics.max_range)


LidarType = IndoorLidar | DualAxisLidar
IntrinsicsType = Indoor8LineLidarIntrinsics | DualAxisLidarIntrinsics


def create_lidar(intrinsics, pose: np.ndarray):
    """Factory keyed on the intrinsics type (reference lidar/indoor_lidar.py:377-393)."""
    if isinstance(intrinsics, DualAxisLidarIntrinsics):
        return DualAxisLidar(intrinsics=intrinsics, pose=pose)
    if isinstance(intrinsics, Indoor8LineLidarIntrinsics):
        return IndoorLidar(intrinsics=intrinsics, pose=pose)
    raise ValueError(f"Unsupported LiDAR intrinsics type: {type(intrinsics)}")


def get_lidar_type(intrinsics) -> str:
    if isinstance(intrinsics, DualAxisLidarIntrinsics):
        return "Dual-axis spiral scanning"
    if isinstance(intrinsics, Indoor8LineLidarIntrinsics):
        if getattr(intrinsics, "dual_axis", False):
            return "Single-axis simulated dual-axis"
        return f"{intrinsics.vertical_res}-line single-axis scanning"
    return "Unknown type"

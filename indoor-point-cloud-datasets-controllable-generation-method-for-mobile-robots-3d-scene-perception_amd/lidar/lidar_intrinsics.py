"""Sensor parameter records (same names, fields and factory values as the reference).

Reference: lidar/lidar_intrinsics.py:12-25 (base), :29-211 (dual axis), :215-389 (multi-line).
Parameter values, factories and accessors; plus the helpers the reference defines without ever calling them
(``calculate_angles_at_time``, ``generate_time_sequence``, ``add_noise``: SURVEY.md F6), kept so that code written
against the reference finds them -- they draw from the global numpy stream in the reference's order
(tests/golden/make_lidar_api_golden.py).
"""
import math

import numpy as np
from abc import ABC
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

_DEG = math.pi / 180.0


@dataclass
class LidarIntrinsics(ABC):
    fov_up: float
    fov_down: float
    vertical_res: int
    horizontal_res: int
    max_range: float
    vertical_degrees: Optional[List[float]] = None


def _even_lines(n: int) -> List[float]:
    """n elevations from +15 to -20 degrees, rounded to 0.1 (lidar_intrinsics.py:273-276, :297-300)."""
    return [round(15.0 - (i * 35.0 / (n - 1.0)), 1) for i in range(n)]


@dataclass
class DualAxisLidarIntrinsics(LidarIntrinsics):
    """BLK2GO-like sensor: 32 lines that each turn 360 deg while nodding +-5 deg."""
    fov_up: float = 15.0
    fov_down: float = 20.0
    vertical_res: int = 1
    horizontal_res: int = 1
    max_range: float = 25.0
    vertical_degrees: Optional[List[float]] = None
    phi_0: float = 0.0
    omega_phi: float = 2.0 * math.pi
    scan_duration: float = 1.0
    point_rate: int = 420000
    phi_range: Tuple[float, float] = (0.0, 2.0 * math.pi)
    theta_range: Tuple[float, float] = (-20.0 * math.pi / 180, 15.0 * math.pi / 180)
    angle_noise_std: float = 0.001
    timing_jitter_std: float = 0.0001
    dropout_probability: float = 0.02
    frame_duration: float = 0.1
    num_vertical_lines: int = 32
    swing_amplitude: float = 5.0 * math.pi / 180
    swing_frequency: float = 1.0

    @classmethod
    def create_blk2go_dual_axis(cls) -> "DualAxisLidarIntrinsics":
        # lidar_intrinsics.py:153-186: 640 kpts/s x 0.1 s = 64 000 rays per scan
        return cls(scan_duration=0.1, point_rate=640000)

    def get_total_points_per_scan(self) -> int:
        return int(self.point_rate * self.scan_duration)

    def get_scan_frequency(self) -> float:
        return 1.0 / self.scan_duration

    def get_range_limits(self) -> tuple:
        return (0.5, self.max_range)

    def get_scan_parameters(self) -> dict:
        keys = ("phi_0", "omega_phi", "scan_duration", "point_rate", "phi_range", "theta_range",
                "swing_amplitude", "swing_frequency")
        return {k: getattr(self, k) for k in keys}

    def swing_angles(self, t, line_idx: int = 0):
        """Noise-free (phi, theta) at time(s) t for one scan line: azimuth turns at omega_phi (mod 2 pi), the
        line's elevation nods around its base angle with a per-line phase, clipped to theta_range
        (lidar_intrinsics.py:82-113 without the noise).  t may be an array."""
        t = np.asarray(t, dtype=np.float64)
        lo, hi = self.theta_range
        L = self.num_vertical_lines
        phi = (self.phi_0 + self.omega_phi * t) % (2 * np.pi)
        base = np.linspace(hi, lo, L)[line_idx % L]
        nod = self.swing_amplitude * np.sin(self.swing_frequency * t + line_idx * 2 * np.pi / L)
        return phi, np.clip(base + nod, lo, hi)

    def calculate_angles_at_time(self, t: float, line_idx: int = 0) -> tuple:
        """(phi, theta) in radians at time t; with angle noise two scalar normals are drawn from the global numpy
        stream, phi's first (lidar_intrinsics.py:82-113)."""
        phi, theta = self.swing_angles(t, line_idx)
        if self.angle_noise_std > 0:
            phi = phi + np.random.normal(0, self.angle_noise_std)
            theta = theta + np.random.normal(0, self.angle_noise_std)
        return phi, theta

    def generate_time_sequence(self, frame_duration: float = None) -> np.ndarray:
        """Sample times of one output frame: int(point_rate * duration) steps of equal length from 0
        (lidar_intrinsics.py:115-135)."""
        span = self.frame_duration if frame_duration is None else frame_duration
        n = int(self.point_rate * span)
        return np.arange(0, span, span / n)

    @classmethod
    def create_custom_dual_axis(cls, phi_0: float = 0.0, theta_0: float = 15.0, omega_phi: float = 2.0 * math.pi,
                                omega_theta: float = -0.1, point_rate: int = 420000, scan_duration: float = 1.0):
        """The reference's factory passes three fields the record does not have (theta_0, omega_theta,
        use_spiral_scan; lidar_intrinsics.py:188-211), so calling it raises TypeError there; it does here too,
        the same way."""
        fields = dict(phi_0=phi_0, theta_0=theta_0 * _DEG, omega_phi=omega_phi, omega_theta=omega_theta,
                      scan_duration=scan_duration, point_rate=point_rate, use_spiral_scan=True, frame_duration=0.1,
                      fov_up=15.0, fov_down=20.0, vertical_res=1, horizontal_res=1, max_range=25.0)
        return cls(**fields)


@dataclass
class Indoor8LineLidarIntrinsics(LidarIntrinsics):
    """Spinning multi-line sensor; 8 lines by default, the factories widen it."""
    fov_up: float = 15.0
    fov_down: float = 20.0
    vertical_res: int = 8
    horizontal_res: int = 2000
    max_range: float = 20.0
    vertical_degrees: Optional[List[float]] = field(
        default_factory=lambda: [15, 10, 5, 0, -5, -10, -15, -20])
    min_range: float = 0.1
    range_resolution: float = 0.01
    scan_frequency: float = 10.0
    points_per_beam: int = 2000
    range_noise_std: float = 0.02
    angle_noise_std: float = 0.01
    dual_axis: bool = False
    capture_rate: int = 200000
    intensity_noise_std: float = 0.1
    dropout_probability: float = 0.05

    @classmethod
    def create_standard_8line(cls):
        return cls()

    @classmethod
    def create_high_resolution_8line(cls):
        return cls(horizontal_res=4000, points_per_beam=4000, range_resolution=0.005)

    @classmethod
    def create_low_cost_8line(cls):
        return cls(horizontal_res=1000, points_per_beam=1000, range_resolution=0.02,
                   range_noise_std=0.05)

    @classmethod
    def create_dense_32line(cls):
        return cls(vertical_res=32, horizontal_res=4000, max_range=25.0,
                   vertical_degrees=_even_lines(32), points_per_beam=3000, range_resolution=0.005,
                   range_noise_std=0.01, angle_noise_std=0.005)

    @classmethod
    def create_leica_blk2go(cls):
        return cls(vertical_res=64, horizontal_res=8000, max_range=25.0,
                   vertical_degrees=_even_lines(64), points_per_beam=5000, range_resolution=0.003,
                   range_noise_std=0.003, angle_noise_std=0.002, min_range=0.5, scan_frequency=20.0,
                   dual_axis=True, capture_rate=420000)

    @classmethod
    def create_custom_lidar(cls, num_beams: int = 8, beam_angles: Optional[List[float]] = None,
                            horizontal_resolution: float = 0.1, max_range: float = 20.0,
                            points_per_beam: int = 2000):
        if beam_angles:
            up, down, degs = max(beam_angles), abs(min(beam_angles)), beam_angles
        else:
            up, down, degs = 15.0, 20.0, [15, 10, 5, 0, -5, -10, -15, -20]
        return cls(fov_up=up, fov_down=down, vertical_res=num_beams,
                   horizontal_res=min(int(360.0 / horizontal_resolution), 10000),
                   max_range=max_range, vertical_degrees=degs, points_per_beam=points_per_beam)

    def get_total_points_per_scan(self) -> int:
        return self.vertical_res * self.horizontal_res

    def get_scan_frequency(self) -> float:
        return self.scan_frequency

    def get_range_limits(self) -> tuple:
        return (self.min_range, self.max_range)

    def add_noise(self, points, ranges, angles, intensities) -> tuple:
        """Measurement noise model the reference declares but never calls (lidar_intrinsics.py:364-389): additive
        normal noise on ranges (range_noise_std), angles (angle_noise_std given in DEGREES) and intensities
        (clipped to [0, 1]), then one uniform draw per point for the dropout.  Global numpy stream, in that order.
        With dropout off the three noisy arrays keep their length and ``points`` is returned as is."""
        ranges, angles, intensities = np.asarray(ranges), np.asarray(angles), np.asarray(intensities)
        r = ranges + np.random.normal(0, self.range_noise_std, ranges.shape)
        a = angles + np.random.normal(0, np.deg2rad(self.angle_noise_std), angles.shape)
        i = np.clip(intensities + np.random.normal(0, self.intensity_noise_std, intensities.shape), 0, 1)
        if self.dropout_probability > 0:
            keep = np.random.random(len(points)) > self.dropout_probability
            return points[keep], r[keep], a[keep], i[keep]
        return points, r, a, i

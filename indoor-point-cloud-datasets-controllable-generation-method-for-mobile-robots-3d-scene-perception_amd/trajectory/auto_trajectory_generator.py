"""Automatic trajectory planning: the pose source of the scan.

Same planner as the reference's AutoTrajectoryGenerator (trajectory/auto_trajectory_generator.py:43-655) -- grid
sampling of the free space at robot height, 2r-connectivity graph, random start/end pairs, A* through the graph,
re-sampling + smoothing, collision counting, scoring -- restated so that, with the same ``np.random`` seed, it
returns the same waypoints bit for bit (tests/golden/planner_golden.npz holds the reference's own outputs).

What changed is where the time went.  The reference decides "is the robot's cube free of mesh vertices?" by a
full pass over all vertices for ONE position at a time: once per grid point and once per waypoint of every
candidate (O(positions x V) numpy passes, :129-139, :345-356).  Here all grid points, and all waypoints of all
candidates, are answered by one HIP kernel launch each (lrc_occ_query).  The O(n^2) Python double loop that
builds the connectivity graph (:245-258) is one vectorised distance matrix.

Float subtleties that decide paths are kept: the A* costs are sums of ``np.linalg.norm`` of point differences, and
ties between equal-length grid paths are broken by the iteration order of Python's ``set``; both are reproduced
by performing the same operations in the same order.
"""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from .trajectory_generator import TrajectoryQuality, Waypoint


@dataclass
class RoomAnalysis:
    bounds: Dict[str, float]
    center: np.ndarray
    dimensions: np.ndarray
    free_space_points: List[np.ndarray]
    obstacle_points: List[np.ndarray]
    connectivity_graph: Dict[int, List[int]]
    mesh: object


@dataclass
class TrajectoryCandidate:
    start_point: np.ndarray
    end_point: np.ndarray
    waypoints: List[Waypoint]
    quality: TrajectoryQuality
    length: float
    collision_count: int
    smoothness_score: float


class AutoTrajectoryGenerator:
    ROBOT_HEIGHT = 1.0

    def __init__(self, robot_radius: float = 0.3, min_trajectory_length: float = None, context=None):
        self.robot_radius = robot_radius
        self.min_trajectory_length = min_trajectory_length
        self.room_analysis: Optional[RoomAnalysis] = None
        self.grid_resolution = 0.2
        self.min_free_space = 1.0
        self.max_candidates = 40
        self.sampling_density = 0.1
        self.interpolation_density = 2.0
        self.min_waypoints = 40
        self._ctx = context
        self._occ = None
        self._occ_key = None

    # ---- robot-cube tests -----------------------------------------------------------------------------
    def _occupancy(self, mesh):
        """Mesh vertices resident on the GPU (one upload per mesh)."""
        v = np.asarray(mesh.vertices)
        # id() alone can be recycled by a new mesh of the same size (a batch loop over scenes), and a mesh can be edited
        # in place: key on the content too -- a full hash of the vertices costs milliseconds next to a plan of seconds
        import hashlib
        key = (id(mesh), v.shape, hashlib.blake2b(np.ascontiguousarray(v).view(np.uint8).reshape(-1).tobytes(),
                                                  digest_size=16).hexdigest())
        if self._occ is None or self._occ_key != key:
            import lidarcast
            if self._ctx is None:
                self._ctx = lidarcast.Context(0)
            if self._occ is not None:
                self._occ.close()
            self._occ = lidarcast.OccupancyIndex(self._ctx, v)
            self._occ_key = key
        return self._occ

    def _in_room_bounds(self, pts: np.ndarray, b: Dict[str, float]) -> np.ndarray:
        """Robot cube completely inside the room box (reference :203-216), for an (n,3) array."""
        lo, hi = pts - self.robot_radius, pts + self.robot_radius
        return ((b["x_min"] <= lo[:, 0]) & (hi[:, 0] <= b["x_max"]) & (b["y_min"] <= lo[:, 1]) &
                (hi[:, 1] <= b["y_max"]) & (b["z_min"] <= lo[:, 2]) & (hi[:, 2] <= b["z_max"]))

    def _blocked(self, pts: np.ndarray, mesh) -> np.ndarray:
        """Robot cube contains a mesh vertex (reference :219-238), all positions in one kernel launch."""
        if len(pts) == 0 or len(np.asarray(mesh.vertices)) == 0:
            return np.zeros(len(pts), dtype=bool)
        return self._occupancy(mesh).occupied(pts, self.robot_radius)

    # the reference's per-position forms of the two tests, and the helpers built on them (:204-243, :404-411)
    def _is_point_in_room_bounds(self, point: np.ndarray, room_bounds: Dict[str, float]) -> bool:
        return bool(self._in_room_bounds(np.asarray(point, dtype=np.float64)[None, :], room_bounds)[0])

    def _is_point_inside_mesh(self, point: np.ndarray, mesh) -> bool:
        return bool(self._blocked(np.asarray(point, dtype=np.float64)[None, :], mesh)[0])

    def _has_sufficient_free_space(self, point: np.ndarray, mesh) -> bool:
        return not self._is_point_inside_mesh(point, mesh)

    @staticmethod
    def _find_nearest_free_space_point(point: np.ndarray, free_space_points: List[np.ndarray]) -> Optional[int]:
        """Index of the closest free-space sample (first one on ties), None without samples."""
        if not free_space_points:
            return None
        return np.argmin([np.linalg.norm(np.array(p) - point) for p in free_space_points])

    # furniture-aware planning: the reference forwards these to its CollisionDetector, which does not import
    # (SURVEY.md F8); out of scope -- furniture inside the room mesh is seen by the cube test
    def add_furniture(self, furniture):
        raise NotImplementedError("furniture-aware collision checking is out of scope; merge furniture into the mesh")

    def add_furniture_from_mesh(self, mesh, name: str, category: str = "unknown"):
        raise NotImplementedError("furniture-aware collision checking is out of scope; merge furniture into the mesh")

    def clear_furniture(self):
        """Nothing is ever registered, so there is nothing to clear."""

    # ---- room analysis --------------------------------------------------------------------------------
    def _sample_free_space(self, mesh, b, resolution):
        xs = np.arange(b["x_min"], b["x_max"], resolution)
        ys = np.arange(b["y_min"], b["y_max"], resolution)
        gx, gy = np.meshgrid(xs, ys, indexing="ij")             # x outer, y inner: the reference's loop order
        pts = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, self.ROBOT_HEIGHT)], axis=1)
        pts = pts[self._in_room_bounds(pts, b)]
        hit = self._blocked(pts, mesh)
        return [p for p in pts[~hit]], [p for p in pts[hit]]

    def _analyze_room_layout(self, mesh, room_bounds) -> RoomAnalysis:
        b = room_bounds
        center = np.array([(b["x_max"] + b["x_min"]) / 2, (b["y_max"] + b["y_min"]) / 2,
                           (b["z_max"] + b["z_min"]) / 2])
        dims = np.array([b["x_max"] - b["x_min"], b["y_max"] - b["y_min"], b["z_max"] - b["z_min"]])
        if self.min_trajectory_length is None:
            self.min_trajectory_length = max(dims[0], dims[1]) * 0.2
        free, blocked = self._sample_free_space(mesh, b, max(0.2, min(dims) / 20))
        if len(free) < 10:                                           # finer second attempt (reference :162-201)
            return self._analyze_room_layout_detailed(mesh, b, center, dims)
        return RoomAnalysis(bounds=b, center=center, dimensions=dims, free_space_points=free,
                            obstacle_points=blocked, connectivity_graph=self._build_connectivity_graph(free),
                            mesh=mesh)

    def _analyze_room_layout_detailed(self, mesh, room_bounds, center: np.ndarray, dimensions: np.ndarray) -> RoomAnalysis:
        """The same analysis on a finer grid, max(0.15, min(dimensions) / 30) (reference :162-201)."""
        free, blocked = self._sample_free_space(mesh, room_bounds, max(0.15, min(dimensions) / 30))
        return RoomAnalysis(bounds=room_bounds, center=center, dimensions=dimensions, free_space_points=free,
                            obstacle_points=blocked, connectivity_graph=self._build_connectivity_graph(free),
                            mesh=mesh)

    def _build_connectivity_graph(self, free: List[np.ndarray]) -> Dict[int, List[int]]:
        """j is a neighbour of i iff |p_i - p_j| <= 2r; neighbour lists in ascending j (reference :245-258)."""
        n = len(free)
        if n == 0:
            return {}
        P = np.asarray(free)
        limit = self.robot_radius * 2
        d = np.sqrt(((P[:, None, :] - P[None, :, :]) ** 2).sum(-1))
        near = d <= limit
        # pairs within rounding distance of the threshold are decided by the reference's own scalar expression
        for i, j in zip(*np.nonzero(np.abs(d - limit) <= 1e-9 * max(limit, 1.0))):
            near[i, j] = np.linalg.norm(free[i] - free[j]) <= limit
        np.fill_diagonal(near, False)
        return {i: np.flatnonzero(near[i]).tolist() for i in range(n)}

    # ---- candidates -----------------------------------------------------------------------------------
    def _draw_endpoint_pairs(self) -> List[Tuple[np.ndarray, np.ndarray]]:
        """The reference's sampling loop (:269-298): two randint draws per attempt, equal or too-close pairs skipped."""
        free = self.room_analysis.free_space_points
        pairs = []
        if len(free) < 2:
            return pairs
        for _ in range(min(self.max_candidates, len(free) * 2)):
            i = np.random.randint(0, len(free))
            j = np.random.randint(0, len(free))
            if i == j:
                continue
            if np.linalg.norm(free[i] - free[j]) < self.min_trajectory_length:
                continue
            pairs.append((free[i], free[j]))
        return pairs

    def _a_star_search(self, start_idx: int, end_idx: int, free: List[np.ndarray]) -> Optional[List[int]]:
        if start_idx == end_idx:
            return [start_idx]
        graph = self.room_analysis.connectivity_graph

        def dist(a, b):
            return np.linalg.norm(free[a] - free[b])

        frontier, done = {start_idx}, set()
        cost = {start_idx: 0.0}
        estimate = {start_idx: dist(start_idx, end_idx)}
        parent = {}
        while frontier:
            cur = min(frontier, key=lambda k: estimate.get(k, float("inf")))
            if cur == end_idx:
                chain = []
                while cur is not None:
                    chain.append(cur)
                    cur = parent.get(cur)
                return chain[::-1]
            frontier.remove(cur)
            done.add(cur)
            for nb in graph.get(cur, []):
                if nb in done:
                    continue
                g = cost[cur] + dist(cur, nb)
                if nb not in frontier:
                    frontier.add(nb)
                elif g >= cost.get(nb, float("inf")):
                    continue
                parent[nb] = cur
                cost[nb] = g
                estimate[nb] = g + dist(nb, end_idx)
        return None

    @staticmethod
    def _generate_linear_waypoints(a, b, n: int) -> List[Waypoint]:
        out = []
        for i in range(n):
            t = i / (n - 1) if n > 1 else 0
            out.append(Waypoint(x=a[0] + t * (b[0] - a[0]), y=a[1] + t * (b[1] - a[1]), z=a[2] + t * (b[2] - a[2]), yaw=0))
        return out

    @staticmethod
    def _generate_waypoints_along_path(path: List[np.ndarray], n: int) -> List[Waypoint]:
        if len(path) < 2:
            return []
        seg = [np.linalg.norm(path[k + 1] - path[k]) for k in range(len(path) - 1)]
        total = 0.0
        for s in seg:
            total += s
        if total < 1e-6:
            p = path[0]
            return [Waypoint(x=p[0], y=p[1], z=p[2], yaw=0)]
        out = []
        for i in range(n):
            if i == n - 1:
                p = path[-1]
                out.append(Waypoint(x=p[0], y=p[1], z=p[2], yaw=0))
                break
            target = (i / (n - 1)) * total
            begin = 0.0
            for k, s in enumerate(seg):
                end = begin + s
                if target <= end:
                    frac = (target - begin) / s if s > 0 else 0
                    p = path[k] + frac * (path[k + 1] - path[k])
                    out.append(Waypoint(x=p[0], y=p[1], z=p[2], yaw=0))
                    break
                begin = end
        return out

    @staticmethod
    def _smooth_trajectory(wps: List[Waypoint], alpha: float = 0.5) -> List[Waypoint]:
        if len(wps) < 3:
            return wps
        out = [wps[0]]
        for i in range(1, len(wps) - 1):
            a, c, b = wps[i - 1], wps[i], wps[i + 1]
            out.append(Waypoint(x=alpha * c.x + (1 - alpha) * (a.x + b.x) / 2,
                                y=alpha * c.y + (1 - alpha) * (a.y + b.y) / 2,
                                z=alpha * c.z + (1 - alpha) * (a.z + b.z) / 2, yaw=c.yaw))
        out.append(wps[-1])
        return out

    def _plan_waypoints(self, start: np.ndarray, end: np.ndarray, n: int) -> List[Waypoint]:
        """Path for one endpoint pair (reference :305-343)."""
        free = self.room_analysis.free_space_points
        if len(free) < 2:
            return self._generate_linear_waypoints(start, end, n)
        P = np.asarray(free)
        si = int(np.argmin(np.sqrt(((P - start) ** 2).sum(1))))
        ei = int(np.argmin(np.sqrt(((P - end) ** 2).sum(1))))
        chain = self._a_star_search(si, ei, free)
        if chain is None or len(chain) < 2:
            return self._generate_linear_waypoints(start, end, n)
        path = [free[k] for k in chain]
        if not np.allclose(path[0], start, atol=0.1):
            path.insert(0, start)
        if not np.allclose(path[-1], end, atol=0.1):
            path.append(end)
        if len(path) == 2:
            return self._generate_linear_waypoints(path[0], path[1], n)
        return self._smooth_trajectory(self._generate_waypoints_along_path(path, n))

    @staticmethod
    def _count_turns(wps: List[Waypoint]) -> int:
        turns = 0
        for i in range(1, len(wps) - 1):
            v1 = np.array([wps[i].x - wps[i - 1].x, wps[i].y - wps[i - 1].y])
            v2 = np.array([wps[i + 1].x - wps[i].x, wps[i + 1].y - wps[i].y])
            n1, n2 = np.linalg.norm(v1), np.linalg.norm(v2)
            if n1 > 1e-6 and n2 > 1e-6:
                ang = np.arccos(np.clip(np.dot(v1 / n1, v2 / n2), -1.0, 1.0))
                if ang > np.pi / 6:
                    turns += 1
        return turns

    @staticmethod
    def _calculate_trajectory_length(wps: List[Waypoint]) -> float:
        total = 0.0
        for i in range(1, len(wps)):
            a, b = wps[i - 1], wps[i]
            total += np.sqrt((b.x - a.x) ** 2 + (b.y - a.y) ** 2 + (b.z - a.z) ** 2)
        return total

    @staticmethod
    def _calculate_smoothness_score(wps: List[Waypoint]) -> float:
        if len(wps) < 3:
            return 1.0
        changes = [abs(wps[i].yaw - wps[i - 1].yaw) for i in range(1, len(wps))]
        return max(0, 1 - np.std(changes) / np.pi)

    def _generate_trajectory_candidates(self, num_waypoints: int) -> List[TrajectoryCandidate]:
        planned = []
        for start, end in self._draw_endpoint_pairs():
            try:
                planned.append((start, end, self._plan_waypoints(start, end, num_waypoints)))
            except Exception:
                continue                                            # the reference drops a failing candidate (:393)
        # collision counts of ALL candidates' waypoints in one launch
        flat = np.array([[w.x, w.y, w.z] for _, _, wps in planned for w in wps], dtype=np.float64).reshape(-1, 3)
        inside = self._in_room_bounds(flat, self.room_analysis.bounds) if len(flat) else np.zeros(0, bool)
        bad = ~inside
        if inside.any():
            bad[inside] = self._blocked(flat[inside], self.room_analysis.mesh)
        out, pos = [], 0
        for start, end, wps in planned:
            n = len(wps)
            collisions = int(bad[pos:pos + n].sum())
            pos += n
            length = self._calculate_trajectory_length(wps)
            smooth = self._calculate_smoothness_score(wps)
            if not wps:
                continue
            quality = TrajectoryQuality(coverage_ratio=1.0 - (collisions / n), path_length=length,
                                        turn_count=self._count_turns(wps),
                                        efficiency=1.0 if collisions == 0 else max(0.0, 1.0 - collisions / n),
                                        collision_count=collisions, smoothness=smooth)
            out.append(TrajectoryCandidate(start_point=start, end_point=end, waypoints=wps, quality=quality,
                                           length=length, collision_count=collisions, smoothness_score=smooth))
        return out

    def _select_best_trajectory(self, candidates: List[TrajectoryCandidate]) -> TrajectoryCandidate:
        if not candidates:
            raise ValueError("No available trajectory candidates")
        best, best_score = None, -1
        for c in candidates:
            score = min(c.length / self.min_trajectory_length, 2.0) * 0.4 + c.smoothness_score * 0.4 \
                - c.collision_count * 0.1
            if score > best_score:
                best, best_score = c, score
        return best

    def _generate_analysis_info(self, candidates, best) -> Dict[str, Any]:
        if not candidates:
            return {}
        lengths = [c.length for c in candidates]
        hits = [c.collision_count for c in candidates]
        smooth = [c.smoothness_score for c in candidates]
        ra = self.room_analysis
        return {
            "total_candidates": len(candidates),
            "best_trajectory": {"length": best.length, "collision_count": best.collision_count,
                                "smoothness_score": best.smoothness_score,
                                "start_point": best.start_point.tolist(), "end_point": best.end_point.tolist()},
            "statistics": {"length_mean": np.mean(lengths), "length_std": np.std(lengths),
                           "collision_mean": np.mean(hits), "collision_std": np.std(hits),
                           "smoothness_mean": np.mean(smooth), "smoothness_std": np.std(smooth)},
            "room_analysis": {"free_space_points": len(ra.free_space_points),
                              "obstacle_points": len(ra.obstacle_points),
                              "room_dimensions": ra.dimensions.tolist(), "room_center": ra.center.tolist()},
        }

    # ---- entry point ----------------------------------------------------------------------------------
    def generate_optimal_trajectory(self, mesh, room_bounds: Dict[str, float],
                                    num_waypoints: int = 20) -> Tuple[List[Waypoint], Dict[str, Any]]:
        self.room_analysis = self._analyze_room_layout(mesh, room_bounds)
        dense = max(int(num_waypoints * self.interpolation_density), self.min_waypoints)
        candidates = self._generate_trajectory_candidates(dense)
        best = self._select_best_trajectory(candidates)
        return best.waypoints, self._generate_analysis_info(candidates, best)

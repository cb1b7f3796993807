"""CPU, world_size 2 and 3 over gloo: pose sharding + the single all-gather assemble the scene cloud
in np.vstack order, independent of the number of ranks.  The per-rank scan is played by the CPU oracle
here (this is the test harness; the product's ranks run the HIP scan)."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scan_block(poses):
    """oracle scan of a block of poses -> (points (K,3) f32, labels (K,) i32, counts (P,))"""
    from lidar import create_lidar
    from lidarcast import synth
    from oracle import np_oracle
    from oracle.c_oracle import OracleMesh
    from helpers import sensor_small
    mesh = synth.make_room(size=(3, 2.5, 2), num_boxes=2, seed=4, cell=0.1)
    om = OracleMesh(mesh.vertices, mesh.triangles).build()
    k = sensor_small(lines=3, width=40, max_range=1.6)
    pts, labs, counts = [], [], []
    for m in poses:
        lidar = create_lidar(k, m)
        p, _, idx = np_oracle.lidar_intersect_mesh(om, lidar, return_index=True)
        _, prim = om.cast(lidar.get_rays())
        lab = mesh.triangle_sem[prim[idx]].astype(np.int32) | (mesh.triangle_ins[prim[idx]].astype(np.int32) << 16)
        pts.append(p)
        labs.append(lab)
        counts.append(len(p))
    return (np.concatenate(pts) if pts else np.zeros((0, 3), np.float32),
            np.concatenate(labs) if labs else np.zeros(0, np.int32), np.array(counts, np.int64), 120)


def _all_poses(n=7):
    from helpers import pose
    return np.stack([pose(0.6 + 0.3 * i, 1.2, 1.0, 0.1 * i) for i in range(n)])


def _worker(rank, world, port, q):
    for p in (PKG, REPO, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from lidarcast.distributed import gather_cloud, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    poses = _all_poses()
    b = shard_bounds(len(poses), world)
    pts, labs, counts, n_per = _scan_block(poses[b[rank]:b[rank + 1]])
    max_local = int(max(b[1:] - b[:-1])) * n_per
    P, L, C = gather_cloud(torch.from_numpy(pts), torch.from_numpy(labs), torch.from_numpy(counts),
                           max_local, dist)
    h = hashlib.sha256(P.numpy().tobytes() + L.numpy().tobytes() + C.numpy().tobytes()).hexdigest()
    q.put((rank, h, int(P.shape[0])))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_cloud_is_rank_count_independent(world):
    pts, labs, counts, _ = _scan_block(_all_poses())
    want = hashlib.sha256(pts.tobytes() + labs.tobytes() + counts.tobytes()).hexdigest()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(pts) > 100
    for rank, h, k in got:
        assert k == len(pts) and h == want, f"rank {rank} assembled a different cloud"


def test_shard_bounds():
    from lidarcast.distributed import shard_bounds
    assert shard_bounds(64, 8).tolist() == list(range(0, 65, 8))
    assert shard_bounds(7, 3).tolist() == [0, 3, 5, 7]
    assert shard_bounds(2, 4).tolist() == [0, 1, 2, 2, 2]
    assert shard_bounds(0, 2).tolist() == [0, 0, 0]


def _pairs_worker(rank, world, port, q):
    """RangeGather protocol over gloo: each rank contributes fixed-stride (t, label) pairs of its pose block;
    rebuilding points from the gathered pairs (numpy stand-in for lrc_cloud_from_ranges_dev, the real kernel is
    checked against the local compaction in the -m gpu tests) gives the vstack cloud on every rank."""
    for p in (PKG, REPO, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from lidarcast.distributed import RangeGather, shard_bounds
    from lidar import create_lidar
    from lidarcast import synth
    from oracle.c_oracle import OracleMesh
    from helpers import sensor_small
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    poses = _all_poses(6)                      # 6 poses: equal blocks for world 2 and 3
    b = shard_bounds(len(poses), world)
    mesh = synth.make_room(size=(3, 2.5, 2), num_boxes=2, seed=4, cell=0.1)
    om = OracleMesh(mesh.vertices, mesh.triangles).build()
    k = sensor_small(lines=3, width=40, max_range=1.6)
    n_per = 120
    mine = poses[b[rank]:b[rank + 1]]
    g = RangeGather(len(mine) * n_per, dist, torch.device("cpu"))
    pairs = np.zeros((len(mine) * n_per, 2), np.int32)
    for j, m in enumerate(mine):
        lidar = create_lidar(k, m)
        rays = lidar.get_rays()
        t, prim = om.cast(rays)
        d = rays[:, 3:] / np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
        pts = rays[:, :3] + d * np.where(np.isfinite(t), t, 0)[:, None]
        keep = np.isfinite(t) & (np.linalg.norm(pts.astype(np.float64) - m[:3, 3], axis=1) < k.max_range)
        t = np.where(keep, t, np.inf).astype(np.float32)
        lab = np.where(keep, mesh.triangle_sem[np.minimum(prim, len(mesh.triangles) - 1)].astype(np.uint32) |
                       (mesh.triangle_ins[np.minimum(prim, len(mesh.triangles) - 1)].astype(np.uint32) << 16), 0)
        pairs[j * n_per:(j + 1) * n_per, 0] = t.view(np.int32)
        pairs[j * n_per:(j + 1) * n_per, 1] = lab.astype(np.uint32).view(np.int32)
    g.slab.copy_(torch.from_numpy(pairs))
    g.gather(async_op=True)
    g.wait()
    allp = g.all_pairs.numpy()
    # rebuild: same formulas as the scan (o + d/|d| * t, float32), global pose order = rank order
    rows = []
    for gp, m in enumerate(poses):
        rays = create_lidar(k, m).get_rays()
        t = allp[gp * n_per:(gp + 1) * n_per, 0].copy().view(np.float32)
        keep = np.isfinite(t)
        d = rays[:, 3:] / np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
        rows.append((rays[:, :3] + d * np.where(keep, t, 0)[:, None])[keep])
    cloud = np.concatenate(rows).astype(np.float32)
    q.put((rank, hashlib.sha256(cloud.tobytes()).hexdigest(), len(cloud)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pair_gather_rebuilds_the_vstack_cloud(world):
    pts, _, _, _ = _scan_block(_all_poses(6))
    want = hashlib.sha256(pts.tobytes()).hexdigest()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pairs_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, h, k in got:
        assert k == len(pts) and h == want, f"rank {rank} rebuilt a different cloud"


def _prims_worker(rank, world, port, q):
    """PrimGather protocol over gloo: each rank contributes the 4-byte hit-triangle ids of its pose block plus
    the per-64-ray keep counts, in one fixed-size slab (a short rank pads with -1 ids / zero counts).  The cloud is
    rebuilt from the gathered ids on every rank (stand-in for lrc_cloud_from_prims_dev, whose kernel the -m gpu
    tests check against the local compaction: t is a function of (ray, triangle), here looked up from the oracle's
    cast of the same ray after checking that it names the same triangle)."""
    for p in (PKG, REPO, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from lidarcast.distributed import PrimGather, shard_bounds
    from lidar import create_lidar
    from lidarcast import synth
    from oracle.c_oracle import OracleMesh
    from helpers import sensor_small
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    poses = _all_poses(7)                      # 7 poses: ragged blocks for world 2 (4+3) and 3 (3+2+2)
    b = shard_bounds(len(poses), world)
    ppr = int(max(b[1:] - b[:-1]))
    mesh = synth.make_room(size=(3, 2.5, 2), num_boxes=2, seed=4, cell=0.1)
    om = OracleMesh(mesh.vertices, mesh.triangles).build()
    k = sensor_small(lines=3, width=64, max_range=1.6)
    n_per = 192

    def scan(m):
        rays = create_lidar(k, m).get_rays()
        t, prim = om.cast(rays)
        d = rays[:, 3:] / np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
        pts = (rays[:, :3] + d * np.where(np.isfinite(t), t, 0)[:, None]).astype(np.float32)
        keep = np.isfinite(t) & (np.linalg.norm(pts.astype(np.float64) - m[:3, 3], axis=1) < k.max_range)
        return pts, prim, keep

    g = PrimGather(ppr, n_per, dist, torch.device("cpu"))
    assert g.fused_counts and g.words % 4 == 0 and g.stride_bytes == g.words * 4
    mine = poses[b[rank]:b[rank + 1]]
    for j, m in enumerate(mine):
        _, prim, keep = scan(m)
        g.prim[j * n_per:(j + 1) * n_per] = torch.from_numpy(np.where(keep, prim, 0xFFFFFFFF).astype(np.uint32)
                                                              .view(np.int32))
        g.tile_count[j * 3:(j + 1) * 3] = torch.from_numpy(keep.reshape(3, 64).sum(1).astype(np.int32))
    g.gather(async_op=True)
    rows, counts = [], []
    for r, (prims, tcs) in enumerate(g.per_rank()):
        for j in range(ppr):
            gp = b[r] + j
            ids = prims[j].view(np.uint32)
            if j >= b[r + 1] - b[r]:                          # padding pose of a short rank
                assert (ids == 0xFFFFFFFF).all() and (tcs[j * 3:(j + 1) * 3] == 0).all()
                continue
            pts, prim, keep = scan(poses[gp])
            sent = ids != 0xFFFFFFFF
            assert np.array_equal(sent, keep) and np.array_equal(ids[sent], prim[keep])
            assert np.array_equal(tcs[j * 3:(j + 1) * 3], sent.reshape(3, 64).sum(1))
            rows.append(pts[sent])
            counts.append(int(sent.sum()))
    cloud = np.concatenate(rows)
    q.put((rank, hashlib.sha256(cloud.tobytes() + np.array(counts, np.int64).tobytes()).hexdigest(), len(cloud)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_triangle_id_gather_rebuilds_the_vstack_cloud(world):
    from helpers import sensor_small
    from lidar import create_lidar
    from lidarcast import synth
    from oracle import np_oracle
    from oracle.c_oracle import OracleMesh
    mesh = synth.make_room(size=(3, 2.5, 2), num_boxes=2, seed=4, cell=0.1)
    om = OracleMesh(mesh.vertices, mesh.triangles).build()
    k = sensor_small(lines=3, width=64, max_range=1.6)
    frames = [np_oracle.lidar_intersect_mesh(om, create_lidar(k, m))[0] for m in _all_poses(7)]
    want = hashlib.sha256(np.concatenate(frames).tobytes()
                          + np.array([len(f) for f in frames], np.int64).tobytes()).hexdigest()
    total = sum(len(f) for f in frames)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prims_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert total > 100
    for rank, h, kk in got:
        assert kk == total and h == want, f"rank {rank} rebuilt a different cloud"


class _StandInEngine:
    """The three device steps of lidarcast.distributed.scan_frames_sharded played on the CPU by the oracle (the
    product's RaycastEngineHIP runs them as HIP kernels; -m gpu tests compare the two).  Everything else -- pose
    sharding, slab padding, the one all-gather, pose-major reassembly, the simulator's frames and statistics -- is
    the product code under test."""

    def __init__(self, mesh):
        from oracle.c_oracle import OracleMesh
        self.mesh = mesh
        self.om = OracleMesh(mesh.vertices, mesh.triangles).build()
        self.ctx = None

    def scene_for(self, mesh):
        return None

    def _direction_table(self, k):
        from lidar import IndoorLidar
        return IndoorLidar.directions_from_vertical_degrees(k.vertical_degrees, k.horizontal_res)

    def _scan(self, k, m):
        from lidar import create_lidar
        rays = create_lidar(k, m).get_rays()
        t, prim = self.om.cast(rays)
        d = rays[:, 3:] / np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
        pts = (rays[:, :3] + d * np.where(np.isfinite(t), t, 0)[:, None]).astype(np.float32)
        keep = np.isfinite(t) & (np.linalg.norm(pts.astype(np.float64) - m[:3, 3], axis=1) < k.max_range)
        return pts, prim, keep

    def prim_gather(self, poses_local, rays_per_pose, dist_, group=None):
        from lidarcast.distributed import PrimGather
        return PrimGather(poses_local, rays_per_pose, dist_, torch.device("cpu"), group=group)

    def scan_block_into(self, g, k, block, mesh):
        n = g.rays_per_pose
        g.slab.fill_(-1)
        g.tile_count.zero_()
        for j, m in enumerate(block.reshape(-1, 4, 4)):
            _, prim, keep = self._scan(k, m)
            g.prim[j * n:(j + 1) * n] = torch.from_numpy(np.where(keep, prim, 0xFFFFFFFF).astype(np.uint32).view(np.int32))
            g.tile_count[j * (n // 64):(j + 1) * (n // 64)] = torch.from_numpy(keep.reshape(-1, 64).sum(1).astype(np.int32))

    def cloud_from_gather(self, g, k, padded_poses, mesh):
        rows, counts = [], []
        per = g.per_rank()
        for r, (prims, _) in enumerate(per):
            for j in range(g.poses_local):
                ids = prims[j].view(np.uint32)
                sent = ids != 0xFFFFFFFF
                counts.append(int(sent.sum()))
                if not sent.any():
                    continue
                pts, prim, keep = self._scan(k, padded_poses[r * g.poses_local + j])
                assert np.array_equal(ids[sent], prim[sent])        # the receiver recomputes the same hit
                lab = (self.mesh.triangle_sem[ids[sent]].astype(np.uint32) |
                       (self.mesh.triangle_ins[ids[sent]].astype(np.uint32) << 16)).view(np.float32)
                rows.append(np.concatenate([pts[sent], lab[:, None]], 1))
        rows = np.concatenate(rows).astype(np.float32) if rows else np.zeros((0, 4), np.float32)
        return rows, np.array(counts, np.int64)

    @staticmethod
    def split_frames(frames, name):
        ends = np.cumsum(frames["counts"])
        return [frames[name][e - c:e] for c, e in zip(frames["counts"], ends)]


def _simulator_worker(rank, world, port, q):
    for p in (PKG, REPO, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank,) + _simulate(distributed=True))
    dist.destroy_process_group()


def _simulate(distributed):
    """S3DISSimulator.run_simulation over 7 yawed poses with the stand-in engine -> (sha of the scene, frame sizes,
    first frame's range_mean).  Not bug-compatible mode, so the incident angles travel too."""
    import s3dis_simulator
    from helpers import sensor_small
    from lidarcast import synth
    from trajectory import Waypoint
    mesh = synth.make_room(size=(3, 2.5, 2), num_boxes=2, seed=4, cell=0.1)
    sim = s3dis_simulator.S3DISSimulator.__new__(s3dis_simulator.S3DISSimulator)
    sim.config, sim.bug_compatible = {}, False
    sim.lidar_config = sensor_small(lines=3, width=64, max_range=1.6)
    sim.raycast_engine = _StandInEngine(mesh)
    from containers import RoomBounds, S3DISScene
    sim.scene = S3DISScene("room", mesh, RoomBounds.from_vertices(mesh.vertices))
    wps = [Waypoint(0.6 + 0.3 * i, 1.2, 1.0, yaw=0.1 * i, timestamp=float(i)) for i in range(7)]
    if distributed:
        scene = sim.run_simulation(wps)                      # finds the initialised process group
    else:                                                   # single process: the stand-in plays scan_frames itself
        import lidarcast.distributed as ld

        class _One:                                          # a world of one rank
            @staticmethod
            def get_world_size(g=None): return 1
            @staticmethod
            def get_rank(g=None): return 0
            @staticmethod
            def all_gather_into_tensor(out, inp, group=None, async_op=False): out.copy_(inp)
        fr = ld.scan_frames_sharded(sim.raycast_engine, sim.lidar_config,
                                    np.stack([w.to_pose_matrix() for w in wps]), mesh, _One)
        return (hashlib.sha256(fr["point3"].tobytes() + fr["sem"].tobytes() + fr["ins"].tobytes()
                               + fr["counts"].tobytes()).hexdigest(), fr["counts"].tolist(), None)
    counts = np.array([len(f.points) for f in scene.frames], np.int64)
    sem, ins = scene.combined_labels()
    h = hashlib.sha256(scene.combined_points().tobytes() + sem.tobytes() + ins.tobytes() + counts.tobytes()).hexdigest()
    ang = np.concatenate([f.incident_angles for f in scene.frames])
    assert ang.dtype == np.float64 and len(ang) == counts.sum() and 0 <= ang.min() and ang.max() <= 90
    return h, counts.tolist(), float(scene.frames[0].scan_quality.range_mean)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_rank_aware_run_simulation(world):
    """S3DISSimulator.run_simulation inside a gloo job of 2, 3, 4 and 8 ranks (ragged pose blocks 4+3 / 3+2+2 / 2+2+2+1 /
    seven ranks with one pose and one with none): every rank
    gets the frames of ALL poses, hash-identical to the single-process assembly, and builds the same statistics."""
    from helpers import sensor_small
    from lidar import create_lidar
    from lidarcast import synth
    from oracle import np_oracle
    from oracle.c_oracle import OracleMesh
    want, want_counts, _ = _simulate(distributed=False)
    mesh = synth.make_room(size=(3, 2.5, 2), num_boxes=2, seed=4, cell=0.1)
    om = OracleMesh(mesh.vertices, mesh.triangles).build()
    k = sensor_small(lines=3, width=64, max_range=1.6)
    from helpers import pose
    frames = [np_oracle.lidar_intersect_mesh(om, create_lidar(k, pose(0.6 + 0.3 * i, 1.2, 1.0, 0.1 * i)))[0]
              for i in range(7)]
    assert want_counts == [len(f) for f in frames] and sum(want_counts) > 100      # = the reference's per-pose loop
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_simulator_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rm = np.linalg.norm(frames[0], axis=1).mean()
    for rank, h, counts, range_mean in got:
        assert counts == want_counts and h == want, f"rank {rank} assembled a different scene"
        assert range_mean == rm

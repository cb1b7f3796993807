#!/usr/bin/env python3
"""tools/pipe_proto.py [scene] -- arrangements of a C3 step (trace + compaction) over internal streams, measured before the
pipeline was written into the library ("the launch tail"): which split of a 64-pose scan into slices, on how many
streams, with which start offsets and stream priorities fills the tail of one launch with the body of the next.
Every arrangement does the same work per step: 64 poses x 65 536 rays traced, 36-byte records written, compacted into
16-byte rows.  (Slices compact into their own row ranges here; the library version shares one scan.)"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lidarcast  # noqa: E402
from lidarcast import synth  # noqa: E402
from lidarcast._capi import LrcCompactIO, LrcHits  # noqa: E402
from lidar import IndoorLidar  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else bench.SCENE
K = int(os.environ.get("PIPE_STEPS", "300"))
mesh = synth.make_scene(name)
ctx = lidarcast.Context(0)
scene = lidarcast.Scene(ctx, mesh.vertices, mesh.triangles, mesh.triangle_sem, mesh.triangle_ins)
sensor = bench.c3_sensor()
poses = bench.c3_poses(0, 1)
P = len(poses)
dirs = IndoorLidar(sensor, np.eye(4)).sensor_directions()
N = len(dirs)
dev = torch.device("cuda", 0)
n = P * N
want = ("t", "prim", "normal3", "point3", "sem", "ins", "tile_count")
sets = [lidarcast.DeviceHits(n, dev, want=want) for _ in range(2)]
clouds = [torch.empty((n, 4), dtype=torch.float32, device=dev) for _ in range(2)]
counts = [torch.zeros(P, dtype=torch.int64, device=dev) for _ in range(2)]
d_poses, d_dirs = torch.from_numpy(poses.reshape(P, 16)).to(dev), torch.from_numpy(dirs).to(dev)
ctxs = [ctx] + [lidarcast.Context(0) for _ in range(3)]          # one compaction scratch per internal stream
ctx.set_launch_chaining(os.environ.get("PIPE_CHAIN", "0") == "1")       # opt-in (off by default)
print("launch chaining (enabled, supported):", ctx.launch_chaining(), flush=True)


class View:
    """poses [a, b) of record set k: an lrc_hits with offset pointers + the compaction of that range into rows [a*N, b*N)"""

    def __init__(self, k, a, b):
        h = sets[k]
        self.a, self.b = a, b
        self.struct = LrcHits()
        for name_, t in h.tensors.items():
            per = t.numel() // (n // 64 if name_ == "tile_count" else n)
            off = a * N // 64 if name_ == "tile_count" else a * N * per
            setattr(self.struct, name_, t.data_ptr() + off * t.element_size())
        io = LrcCompactIO()
        io.t, io.point3 = self.struct.t, self.struct.point3
        io.sem, io.ins, io.tile_count = self.struct.sem, self.struct.ins, self.struct.tile_count
        io.counts = counts[k].data_ptr() + a * 8
        io.out_xyzl = clouds[k].data_ptr() + a * N * 16
        self.io = io
        self.poses = d_poses[a:b]


def trace(v, stream):
    scene.scan_poses_dev(v.poses, d_dirs, v, sensor.max_range, stream.cuda_stream)


def compact(v, stream, c):
    ctxs[c].compact_dev(v.b - v.a, N, v.io, stream.cuda_stream)


ONLY = os.environ.get("PIPE_ONLY")


def timed(label, body, prologue=None):
    if ONLY and ONLY not in label:
        return 0.0
    for rep in range(2):
        torch.cuda.synchronize()
        if prologue:
            prologue()
        t0 = time.perf_counter()
        for i in range(K):
            body(i)
        t_sub = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{label:58s} {dt / K * 1e3:.4f} ms/step = {n * K / dt / 1e9:.2f} G rays/s   (host submit {t_sub / K * 1e3:.3f} ms/step)", flush=True)
    return dt / K


def streams(k, prios=None):
    return [torch.cuda.Stream(device=dev, priority=(prios[i] if prios else 0)) for i in range(k)]


# ---- 0. the bench's step: one stream, one launch, compaction behind it
s0 = streams(1)
full = [View(0, 0, P), View(1, 0, P)]
base = timed("serial: trace(64) + compact, one stream", lambda i: (trace(full[0], s0[0]), compact(full[0], s0[0], 0)))

# ---- 1. whole scans alternating on two streams, two record sets (a caller with two launches in flight)
s2 = streams(2)
timed("whole scans alternating on 2 streams, 2 record sets",
      lambda i: (trace(full[i % 2], s2[i % 2]), compact(full[i % 2], s2[i % 2], i % 2)))


def slices(k, parts):
    """parts: list of pose ranges"""
    return [View(k, a, b) for a, b in parts]


def halves(h=P // 2):
    return [(0, h), (h, P)]


# ---- 2. two half-scan chains, one record set, no offset (both chains start together)
def chains(parts_per_stream, nstreams, prios=None, offset=None, label=""):
    ss = streams(nstreams, prios)
    vs = [slices(0, parts) for parts in parts_per_stream]

    def body(i):
        for c, vlist in enumerate(vs):
            for v in vlist:
                trace(v, ss[c])
                compact(v, ss[c], c)

    def prologue():
        if offset:
            # a one-time extra launch on chain 0: shifts its boundaries by a fraction of a slice for the whole run
            trace(View(1, 0, offset), ss[0])
    return timed(label, body, prologue)


chains([[(0, 32)], [(32, 64)]], 2, label="2 chains x half scans, in phase")
for off in (8, 16, 24):
    chains([[(0, 32)], [(32, 64)]], 2, offset=off, label=f"2 chains x half scans, chain 0 offset by {off} poses once")
chains([[(0, 32)], [(32, 64)]], 2, prios=[-1, 0], label="2 chains x half scans, chain 0 high priority")
chains([[(0, 32)], [(32, 64)]], 2, prios=[-1, 0], offset=16, label="2 chains x half scans, high priority + offset 16")
chains([[(0, 24)], [(24, 64)]], 2, label="2 chains, uneven 24 / 40")
chains([[(0, 16), (16, 32)], [(32, 48), (48, 64)]], 2, label="2 chains x 2 quarter scans each, in phase")
chains([[(0, 16), (16, 32)], [(32, 48), (48, 64)]], 2, offset=8, label="2 chains x 2 quarter scans each, offset 8")
chains([[(0, 16)], [(16, 32)], [(32, 48)], [(48, 64)]], 4, label="4 chains x quarter scans, in phase")
chains([[(0, 22)], [(22, 43)], [(43, 64)]], 3, label="3 chains x third scans, in phase")


# ---- 3. round robin: slice j of the endless sequence on stream j % S, each stream a serial chain
def round_robin(nslices, nstreams, label, offset=None):
    ss = streams(nstreams)
    step = P // nslices
    vs = [View(0, j * step, (j + 1) * step) for j in range(nslices)]
    st = {"j": 0}

    def body(i):
        for v in vs:
            c = st["j"] % nstreams
            st["j"] += 1
            # a slice's records are rewritten one step later on whatever stream it lands: order it after its own compaction
            trace(v, ss[c])
            compact(v, ss[c], c)

    def prologue():
        st["j"] = 0
        if offset:
            trace(View(1, 0, offset), ss[0])
    if nslices % nstreams:
        return None          # a slice would change streams from step to step: needs events, not measured here
    return timed(label, body, prologue)


round_robin(4, 2, "4 quarter slices round robin on 2 streams")
round_robin(4, 2, "4 quarter slices round robin on 2 streams, offset 8", offset=8)
round_robin(8, 2, "8 slices round robin on 2 streams, offset 4", offset=4)
print(f"baseline serial step {base * 1e3:.4f} ms")


# ---- 4. trace chains free of the compaction: two record sets, compactions on a third stream --------------------------
def free_chains(parts, nchains, label, compaction=True, csplit=1, prios=None):
    """parts: pose ranges, slice j on trace stream j % nchains; step i uses record set i % 2; the compaction of a slice runs on
    the side stream(s) after the slice's trace; a slice's next-but-one trace (same record set) waits for that compaction."""
    ts = streams(nchains, prios)
    cs = streams(csplit)
    vs = [[View(k, a, b) for (a, b) in parts] for k in range(2)]
    done = [[None] * len(parts) for _ in range(2)]        # event: compaction of (set, slice) finished

    def body(i):
        k = i % 2
        for j, v in enumerate(vs[k]):
            t = ts[j % nchains]
            if compaction and done[k][j] is not None:
                t.wait_event(done[k][j])
            trace(v, t)
            if compaction:
                ev = torch.cuda.Event()
                ev.record(t)
                c = cs[j % csplit]
                c.wait_event(ev)
                compact(v, c, 2 + (j % csplit))
                ev2 = torch.cuda.Event()
                ev2.record(c)
                done[k][j] = ev2

    def prologue():
        for k in range(2):
            for j in range(len(parts)):
                done[k][j] = None
    return timed(label, body, prologue)


free_chains([(0, 64)], 1, "traces only: whole scans, one stream", compaction=False)
free_chains([(0, 64), (0, 64)], 2, "traces only: whole scans alternating on 2 streams (2 scans per step!)", compaction=False)
free_chains([(0, 32)], 1, "traces only: HALF scan, one stream (half the rays per step!)", compaction=False)
free_chains([(0, 32), (32, 64)], 2, "traces only: half scans on 2 chains", compaction=False)
free_chains([(0, 64)], 1, "whole scans on 1 trace stream, compaction on a side stream")
free_chains([(0, 32), (32, 64)], 2, "half scans on 2 trace chains, compaction on a side stream")
free_chains([(0, 32), (32, 64)], 2, "half scans on 2 trace chains, compaction on 2 side streams", csplit=2)
free_chains([(0, 32), (32, 64)], 2, "half scans on 2 chains (high prio), compaction side (low)", prios=[-1, -1])
free_chains([(0, 22), (22, 43), (43, 64)], 3, "third scans on 3 trace chains, compaction on a side stream")
free_chains([(0, 16), (16, 32), (32, 48), (48, 64)], 2, "quarter scans on 2 trace chains, compaction on a side stream")
free_chains([(0, 16), (16, 32), (32, 48), (48, 64)], 4, "quarter scans on 4 trace chains, compaction on a side stream")

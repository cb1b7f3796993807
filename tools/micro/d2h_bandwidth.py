#!/usr/bin/env python3
"""tools/micro/d2h_bandwidth.py -- device -> page-locked host copies of the C3 result size (67 MB): one stream against
two and four streams each taking a part, and against chunk sizes."""
import time
import torch

dev = torch.device("cuda", 0)
n = 67 * 1024 * 1024
src = torch.empty(n, dtype=torch.uint8, device=dev)
dst = torch.empty(n, dtype=torch.uint8).pin_memory()
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]


def run(parts, nstreams, reps=20):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        step = n // parts
        for k in range(parts):
            with torch.cuda.stream(streams[k % nstreams]):
                dst[k * step:(k + 1) * step].copy_(src[k * step:(k + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for parts, ns in ((1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 2), (8, 4), (16, 4)):
    t = run(parts, ns)
    print(f"{parts:2d} parts on {ns} stream(s): {t * 1e3:.3f} ms = {n / t / 1e9:.1f} GB/s")

#!/usr/bin/env python3
"""tools/vw_timeline.py <kernel_trace.csv> -- start/end/duration of the trace and rebuild kernels of a
bench.py --dist-selftest run under rocprofv3 --kernel-trace (who overlaps whom, what each costs alone)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows
      if any(k in r["Kernel_Name"] for k in ("trace_kernel", "prim_scatter", "cloud_scatter", "compact_scatter"))]
ks.sort()
t0 = ks[0][0]
alone, over = {}, {}
for j, (a, b, n) in enumerate(ks):
    name = next(k for k in ("trace_kernel", "prim_scatter", "cloud_scatter", "compact_scatter") if k in n)
    overlapped = any(c < b and d > a for i, (c, d, _) in enumerate(ks) if i != j)
    (over if overlapped else alone).setdefault(name, []).append((b - a) / 1e3)
for tag, d in (("alone", alone), ("overlapped", over)):
    for name, v in sorted(d.items()):
        v.sort()
        print(f"{tag:10s} {name:26s} n={len(v):3d} median={v[len(v)//2]:8.1f} us  min={v[0]:8.1f}  max={v[-1]:8.1f}")

#!/usr/bin/env python3
"""tools/kernel_timeline.py <rocprofv3 kernel_trace.csv> [first_row] [rows] -- begin / end of consecutive kernel launches
relative to the first one shown, one line per launch, with the queue it ran on: which launches overlap which."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
count = int(sys.argv[3]) if len(sys.argv) > 3 else 24
sel = rows[first:first + count]
t0 = int(sel[0]["Start_Timestamp"])
short = {"trace_kernel": "trace", "compact_scan": "scan", "compact_scatter": "scatter", "prim_scatter": "rebuild"}
for r in sel:
    name = r["Kernel_Name"]
    for k, v in short.items():
        if k in name:
            name = v
            break
    b, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r.get("Queue_Id", "?")
    grid = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    print(f"queue {q:>3}  {name[:28]:28s} grid {grid:>9}  start {b:9.1f} us  end {e:9.1f} us  ({e - b:7.1f} us)")

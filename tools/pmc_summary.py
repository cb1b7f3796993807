#!/usr/bin/env python3
"""Average each PMC counter per dispatch of every kernel whose name matches (default: trace_kernel)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        per[(r["Dispatch_Id"], r["Counter_Name"], r["Kernel_Name"][:60])] += float(r["Counter_Value"])
    for (d, c, k), v in per.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    print(k)
    for c, vs in sorted(cs.items()):
        print(f"  {c:34s} n={len(vs):3d} mean={sum(vs)/len(vs):.6g}")

# counter summary of the pose-batched trace kernel for bench.py (profiles/pmc_latest.json), stamped with the
# fingerprint of the sources the profiled binary was built from: bench.py ignores it when the tree has moved on
import json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
for k, cs in acc.items():
    if ("trace_kernel<1" in k or "trace_refill" in k) and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs and "SQ_INSTS_VALU" in cs:
        mean = lambda v: sum(v) / len(v)
        out = {"kernel": k, "source_sha256": entry.source_fingerprint(),
               "FETCH_SIZE_KB": mean(cs["FETCH_SIZE"]), "WRITE_SIZE_KB": mean(cs["WRITE_SIZE"]),
               "rays_per_launch": 4194304, "scene": os.environ.get("PMC_SCENE", "synth_A6_office2"),
               "command": "bench.py " + os.environ.get("PMC_ARGS", "--steps 3 --warmup 1 --no-cpu-baseline --no-caller-path"),
               "counters": {c: mean(v) for c, v in sorted(cs.items())}}
        with open(os.path.join(root, "pmc.json"), "w") as f:
            json.dump(out, f, indent=1)

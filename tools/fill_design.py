#!/usr/bin/env python3
"""tools/fill_design.py -- fills the {{PLACEHOLDERS}} of tools/design_front.md from the committed evidence (profiles/r04_bench.json,
r04_kernel_stats_serial.csv, r04_build_kernel_stats.csv, pmc_latest.json) and composes DESIGN.md (tools/compose_design.py); the
template itself stays as it is.  One figure per quantity, all from the files the table cites."""
import csv
import json
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(R)
d = json.loads(open("profiles/r04_bench.json").read().strip().splitlines()[-1])
c, r, cpu = d["config"], d["roofline"], d["cpu_baseline"]
rows = list(csv.DictReader(open("profiles/r04_kernel_stats_serial.csv")))
trace = [x for x in rows if "trace_kernel" in x["Name"]][0]
b = list(csv.DictReader(open("profiles/r04_build_kernel_stats.csv")))
builds = [int(x["Calls"]) for x in b if "k_subtree" in x["Name"]][0]
launches = sum(int(x["Calls"]) for x in b) / builds
pmc = json.load(open("profiles/pmc_latest.json"))
vw = open("profiles/r04_multigpu_virtual_world.txt").read()
vals = {
    "MS_PER_STEP": f"{d['ms_per_step']:.3f}", "VALUE_G": f"{d['value'] / 1e9:.2f}", "SERIAL_MS": f"{c['serial_ms_per_step']:.3f}",
    "KERNEL_MS": f"{r['kernel_ms']:.3f}", "ROCPROF_MS": f"{float(trace['AverageNs']) / 1e6:.3f}",
    "FRAC": f"{r['frac']:.3f}", "ISSUE": f"{r['valu_issue_frac']:.3f}", "LANES": f"{r['lane_utilisation']:.3f}",
    "TRAFFIC_GB": f"{r['traffic'] / 1e9:.2f}", "SCENE_CREATE": f"{c['scene_create_ms']:.2f}",
    "UPLOAD_MS": f"{c['bvh']['transfer_ms']:.2f}", "BUILD_MS": f"{c['bvh']['build_ms']:.2f}",
    "TRAJ_G": f"{c['trajectory_including_scene_create_rays_per_s'] / 1e9:.2f}", "CALLER_MS": f"{c['caller_path_ms']:.2f}",
    "RUN_SIM_MS": f"{c['run_simulation_ms']:.2f}", "CPU_VALUE": f"{cpu['value'] / 1e6:.2f}",
    "CPU_ONCE": f"{cpu['build_once_value'] / 1e6:.2f}", "BUILD_LAUNCHES": f"{launches:.0f}",
    "VW": os.environ.get("VW", "see `profiles/r04_multigpu_virtual_world.txt`"),
}
front = open("tools/design_front.md").read()
filled = front
for k, v in vals.items():
    filled = filled.replace("{{" + k + "}}", v)
left = [k for k in vals if "{{" + k + "}}" in filled]
assert not left, left
shutil.copy("tools/design_front.md", "/tmp/design_front_template.md")
try:
    open("tools/design_front.md", "w").write(filled)
    subprocess.check_call([sys.executable, "tools/compose_design.py", "766412b"])
finally:
    shutil.copy("/tmp/design_front_template.md", "tools/design_front.md")
print({k: vals[k] for k in ("MS_PER_STEP", "VALUE_G", "FRAC", "SCENE_CREATE", "BUILD_LAUNCHES", "RUN_SIM_MS", "CPU_VALUE")})

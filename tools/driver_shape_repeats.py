#!/usr/bin/env python3
"""Runs the driver's command shape (`bench.py --steps 20 --warmup 5`, step leg only) N times on this box and prints the spread of
the region's clocks: how much of a difference between two bench lines is the box and how much the code.  Usage: tools/driver_shape_repeats.py [N]"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rows = []
for _ in range(n):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-vector-env", "--no-cpu-baseline", "--rollout", "0"],
                         capture_output=True, text=True, check=True).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1]); r = d["roofline"]
    rows.append({"value": d["value"], "wall_us": d["ms_per_step"] * 20e3, "device_region_us": r["device_region_us"], "host_overhead_us": r["host_overhead_us"],
                 "frac": r["frac"], "frac_device": r["frac_device"], "launch_call_us": r["host_timeline_us"]["launch_call"]})
summary = {k: {"min": min(x[k] for x in rows), "median": statistics.median(x[k] for x in rows), "max": max(x[k] for x in rows)} for k in rows[0]}
print(json.dumps({"runs": n, "command": "bench.py --steps 20 --warmup 5 --no-vector-env --no-cpu-baseline --rollout 0", "summary": summary, "rows": rows}))

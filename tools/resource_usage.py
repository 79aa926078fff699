#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (build/asm/resource_usage.txt)."""
import re, subprocess, sys
path = sys.argv[1] if len(sys.argv) > 1 else "build/asm/resource_usage.txt"
rows, cur = [], {}
for line in open(path):
    m = re.search(r"remark: [^:]+:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        if cur: rows.append(cur)
        name = body.split(":", 1)[1].strip()
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
        cur = {"name": re.sub(r"\(.*", "", name)}
    elif ":" in body:
        k, v = body.split(":", 1); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
print("%-62s %5s %5s %5s %7s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ"))
for r in rows:
    print("%-62s %5s %5s %5s %7s %6s" % (r["name"][-62:], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
          r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))

#!/usr/bin/env python3
"""Joins the kernel names a default-schedule walk launched (rocprofv3 kernel_stats.csv of tools/schedule_walk.py) with the per-kernel
register / scratch table of the build (tools/spill_sites.py + hipcc -Rpass-analysis): which default-schedule kernels carry scratch, and
whether any of it sits inside a loop.  usage: kernel_scratch_report.py <kernel_stats.csv>   (host only apart from the csv)"""
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n[:n.index("(")] if "(" in n else n
    return re.sub(r"\s+", "", n)


launched = {}
for r in csv.DictReader(open(sys.argv[1])):
    launched[norm(r["Name"])] = launched.get(norm(r["Name"]), 0) + int(r["Calls"])
spill = {}
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "spill_sites.py")], capture_output=True, text=True).stdout
for ln in out.splitlines()[1:]:
    m = re.match(r"^(.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)$", ln)
    if m:
        spill[norm(m.group(1))] = tuple(int(x) for x in m.groups()[1:])
ours = {k: v for k, v in launched.items() if k.startswith("lh::")}
print(f"# {len(ours)} distinct library kernels launched by the default-schedule walk (tools/schedule_walk.py: 7B / 13B / 30B / 65B widths, every entry point, no environment switch)")
print(f"# {sum(1 for k in ours if k in spill)} of them carry scratch; {sum(1 for k in ours if k in spill and spill[k][2])} with a spill instruction inside a loop")
print("%-60s %8s %8s %6s %9s %9s" % ("kernel", "calls", "scratch", "VGPRs", "in loops", "outside"))
for k in sorted(ours, key=lambda k: (-(spill.get(k, (0,))[0]), k)):
    s = spill.get(k)
    print("%-60s %8d %8s %6s %9s %9s" % (k, ours[k], s[0] if s else 0, s[1] if s else "", s[2] if s else "", s[3] if s else ""))

#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats --output-format csv` run into a small text table
(what gets committed under profiles/).  usage: prof_summary.py <*_kernel_stats.csv> [title]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
title = sys.argv[2] if len(sys.argv) > 2 else ""
if title:
    print("#", title)
print(f"{'calls':>8} {'avg_us':>9} {'min_us':>8} {'max_us':>8} {'total_ms':>9} {'pct':>6}  kernel")
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "")
    n = n[:n.index("(")] if "(" in n else n
    print(f"{int(r['Calls']):8d} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:8.2f} {float(r['MaxNs'])/1e3:8.2f} "
          f"{float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):6.2f}  {n}")

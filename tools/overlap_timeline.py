#!/usr/bin/env python3
"""Timeline of the decode launches from a rocprofv3 --kernel-trace CSV (measurement tooling): for a window of tokens in the middle
of the trace, every dispatch with its start / end relative to the token's first launch, its queue, and how much of it overlapped
the previous dispatch; then the per-layer period (end of w2 to end of w2) and the union / sum of the kernel intervals.
usage: overlap_timeline.py <dir-or-kernel_trace.csv> [--token N] [--layers 3]"""
import argparse
import csv
import glob
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("--token", type=int, default=-1, help="index of the token to print (default: the middle one)")
ap.add_argument("--layers", type=int, default=3, help="layers of that token to print in full")
args = ap.parse_args()
files = [args.src] if os.path.isfile(args.src) else glob.glob(os.path.join(args.src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        low = {k.lower(): v for k, v in r.items()}
        try:
            rows.append((int(low["start_timestamp"]), int(low["end_timestamp"]), low.get("kernel_name", ""), low.get("queue_id", "?")))
        except (KeyError, ValueError):
            pass
rows.sort()
if not rows:
    sys.exit("no dispatches found")


def short(n):
    n = n[:n.index("(")] if "(" in n else n
    return n.replace("void ", "").replace("lh::", "")


# tokens end with k_argmax
ends = [i for i, r in enumerate(rows) if "k_argmax" in r[2]]
if len(ends) < 3:
    sys.exit("fewer than 3 decode tokens in the trace")
ti = args.token if args.token >= 0 else len(ends) // 2
lo, hi = ends[ti - 1] + 1, ends[ti] + 1
tok = rows[lo:hi]
t0 = tok[0][0]
print(f"# token {ti} of {len(ends)}: {len(tok)} dispatches, span {(tok[-1][1] - t0) / 1e3:.1f} us; sum of kernel durations {sum(r[1] - r[0] for r in tok) / 1e3:.1f} us")
# union of intervals
iv = sorted((r[0], r[1]) for r in tok)
un, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        un += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
un += ce - cs
print(f"# union of kernel intervals {un / 1e3:.1f} us (span - union = idle: {((tok[-1][1] - t0) - un) / 1e3:.1f} us)")
n_show = 2 + 4 * args.layers
print(f"{'start':>9} {'end':>9} {'dur':>8} {'ovl_prev':>8}  queue  kernel")
prev_end = None
for k, r in enumerate(tok):
    if k < n_show or k >= len(tok) - 3:
        ovl = max(0, min(prev_end, r[1]) - r[0]) if prev_end is not None else 0
        print(f"{(r[0] - t0) / 1e3:9.2f} {(r[1] - t0) / 1e3:9.2f} {(r[1] - r[0]) / 1e3:8.2f} {ovl / 1e3:8.2f}  {r[3]:>5}  {short(r[2])}")
    elif k == n_show:
        print("      ...")
    prev_end = r[1] if prev_end is None else max(prev_end, r[1])
# per-layer period over all tokens: ends of the resid-role launch that closes a layer = every 4th k_gemv after k_qkv_attn
periods = []
for a in range(1, len(ends)):
    seg = rows[ends[a - 1] + 1:ends[a] + 1]
    qk = [i for i, r in enumerate(seg) if "k_qkv_attn" in r[2]]
    if len(qk) < 3:
        continue
    starts = [seg[i][0] for i in qk]
    periods += [(b - a_) / 1e3 for a_, b in zip(starts[:-1], starts[1:])]
if periods:
    periods.sort()
    print(f"# layer period (start of k_qkv_attn to the next one), all tokens: median {periods[len(periods) // 2]:.2f} us, mean {sum(periods) / len(periods):.2f} us, n = {len(periods)}")

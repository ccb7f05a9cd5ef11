#!/usr/bin/env python3
"""Where the register spills of each kernel sit: for every kernel of the library's HIP translation units whose ScratchSize is not 0,
the scratch_load / scratch_store instructions INSIDE a loop (between a label and a backward branch to it) against those outside
(prologue / epilogue code that runs once per workgroup).  Host only: hipcc -S for gfx950, no device needed.
usage: tools/spill_sites.py [file.hip ...]          (default: every .hip under llama.swift_amd/csrc)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "llama.swift_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
FLAGS = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S"]


def demangle(names):
    return subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")


print("%-58s %7s %8s %8s %9s %9s" % ("kernel", "scratch", "VGPRs", "in loops", "outside", "loops"))
for f in files:
    out = f"/tmp/spill_{os.path.basename(f)}.s"
    r = subprocess.run(FLAGS + ["-o", out, os.path.join(CSRC, os.path.basename(f))], capture_output=True, text=True)
    if r.returncode != 0:
        print(f"# {f}: compile failed\n{r.stderr[-500:]}")
        continue
    lines = open(out).read().split("\n")
    # kernel bodies: "<name>:" ... ".end_amdhsa_kernel" metadata follows; scratch size from the .amdhsa_private_segment_fixed_size line
    starts = [(i, m.group(1)) for i, ln in enumerate(lines) for m in [re.match(r"^(_Z\w+):\s", ln)] if m]
    kern = []
    for idx, (i, name) in enumerate(starts):
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        body = lines[i:end]
        scratch = next((int(m.group(1)) for ln in body for m in [re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", ln)] if m), None)
        vgpr = next((int(m.group(1)) for ln in body for m in [re.search(r"; NumVgprs: (\d+)", ln)] if m), None)
        if scratch is None:
            continue
        kern.append((name, scratch, vgpr, body))
    names = demangle([k[0] for k in kern])
    for (name, scratch, vgpr, body), dn in zip(kern, names):
        if not scratch:
            continue
        stop = next((j for j, ln in enumerate(body) if ln.strip().startswith(".section") or ".end_amdhsa_kernel" in ln), len(body))
        code = body[:stop]
        labels = {m.group(1): j for j, ln in enumerate(code) for m in [re.match(r"^(\.LBB\w+):", ln)] if m}
        loops = []
        for j, ln in enumerate(code):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", ln)
            if m:
                t = labels.get(m.group(1) or m.group(2))
                if t is not None and t <= j:
                    loops.append((t, j))
        inside = outside = 0
        for j, ln in enumerate(code):
            if re.search(r"\bscratch_(load|store)", ln):
                if any(a <= j <= b for a, b in loops):
                    inside += 1
                else:
                    outside += 1
        dn = re.sub(r"\(.*", "", dn).replace("void ", "")
        print("%-58s %7d %8s %8d %9d %9d" % (dn, scratch, vgpr, inside, outside, len(loops)))

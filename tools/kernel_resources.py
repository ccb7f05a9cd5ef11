#!/usr/bin/env python3
"""Per-kernel VGPR / occupancy / scratch table of one HIP translation unit (default decode.hip; second argument: prep.hip | prompt_gemm.hip | prompt_attn.hip) (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py [substring-of-demangled-name]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "llama.swift_amd", "csrc", (sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].endswith(".hip") else "decode.hip"))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
extra = [a for a in sys.argv[2:] if not a.endswith(".hip")]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src,
                    "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
for b, d in zip(blocks, dem):
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    if flt not in d:
        continue
    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-48s VGPR %4s AGPR %3s SGPR %3s occ %s scratch %s lds %s" % (d, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("Occupancy [waves/SIMD]"),
                                                                  g("ScratchSize [bytes/lane]"), g("LDS Size [bytes/block]")))

#!/usr/bin/env python3
"""Summarises the three rocprofv3 --pmc passes over tools/gemv_probe.py into profiles/ (text table +
gemv_traffic.json used by bench.py for roofline.traffic).
usage: pmc_summary.py <pass1_counter_collection.csv> <pass2 FETCH_SIZE csv> <pass3 WRITE_SIZE csv> <out.txt> <out.json>"""
import collections
import csv
import json
import sys

SHAPES = {  # (template D/RING substring, grid threads) -> (name, M, K)
    98304: ("wq|wk|wv", 12288, 4096), 176128: ("w1|w3", 22016, 4096), 256000: ("output", 32000, 4096),
}


def key_of(r):
    k = r["Kernel_Name"]
    if "k_gemv" not in k:
        return None
    grid = int(r["Grid_Size"])
    if grid in SHAPES:
        return SHAPES[grid]
    if grid == 32768:
        return ("wo", 4096, 4096) if "16, false" in k else ("w2", 4096, 11008)
    return None


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = key_of(r)
        if k:
            a[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in a.items()}


q, f, w = agg(sys.argv[1]), agg(sys.argv[2]), agg(sys.argv[3])
lines = [
    "# rocprofv3 --pmc passes over tools/gemv_probe.py: the decode GEMV kernel lh::k_gemv<PRE_QA, STORE> on every matrix kind of the",
    "# synthetic LLaMA-7B model, launches cycling through all 32 layers (cold weights) -- MI355X, the kernels of the build the file name says.",
    "# pass 1: --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE",
    "# pass 2: --pmc FETCH_SIZE        pass 3: --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum        (separate passes, --kernel-trace only)",
    "# FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read",
    "# (MI355X_MICROARCH.md, HBM section): traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.  SQ_* are quad-cycles summed over all waves.",
    f"{'shape':>10} {'M':>6} {'K':>6} {'algo_MB':>9} {'traffic_MB':>11} {'ratio':>6} | {'wave_cyc':>10} {'wait_any%':>9} {'wait_inst%':>10} {'active%':>8} {'valu/wave':>9}",
]
out = {}
for k in sorted(f, key=lambda x: x[1] * x[2]):
    name, M, K = k
    algo = M * (K // 32) * 20 + (K // 32) * 20 + 4 * M
    traffic = 2 * f[k]["FETCH_SIZE"] * 1024 + w[k]["WRITE_SIZE"] * 1024
    c = q[k]
    wc = c["SQ_WAVE_CYCLES"]
    waves = (M + 7) // 8
    lines.append(f"{name:>10} {M:6d} {K:6d} {algo / 1e6:9.2f} {traffic / 1e6:11.2f} {traffic / algo:6.3f} | {wc:10.0f} "
                 f"{100 * c['SQ_WAIT_ANY'] / wc:9.1f} {100 * c['SQ_WAIT_INST_ANY'] / wc:10.1f} {100 * c['SQ_ACTIVE_INST_ANY'] / wc:8.1f} {c['SQ_INSTS_VALU'] / waves:9.0f}")
    out[name] = {"M": M, "K": K, "traffic_bytes": traffic, "algo_bytes": algo}
open(sys.argv[4], "w").write("\n".join(lines) + "\n")
json.dump({"source": sys.argv[4] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 FETCH_SIZE correction)", "per_launch": out},
          open(sys.argv[5], "w"), indent=1)
print("\n".join(lines))

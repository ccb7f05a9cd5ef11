#!/usr/bin/env python3
"""GEMV kernel probe for rocprofv3 runs: loads the synthetic 7B model and launches only the decode
GEMV kernel (PRE_QA / STORE) on each matrix kind, cycling through all layers (cold weights) and on
one layer repeatedly (Infinity-Cache resident).  usage: gemv_probe.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L  # noqa: E402

path = os.path.join(os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models"), "7B-seed20230312", "ggml-model-q4_0.bin")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m = L.Model(path, n_ctx=512)
for which in range(5):
    a = m.bench_gemv(which, -1, 1, iters)
    b = m.bench_gemv(which, 0, 3, iters * 8)
    print(f"{a['name']:9s} M={a['M']:6d} K={a['K']:6d}  cold {a['us_per_launch']:7.2f} us {a['GBps']:7.0f} GB/s   "
          f"cache-resident {b['us_per_launch']:7.2f} us {b['GBps']:7.0f} GB/s")
# prologue / epilogue ablation of the two norm-fused matrices (variant << 4 | which)
import ctypes as C
from llama_swift_amd.binding import _GemvBench
for which, name in ((2, "w1|w3"), (0, "wq|wk|wv")):
    row = []
    for variant, label in ((0, "QA/STORE"), (1, "NORM/STORE"), (3, "QA/SILU_QA"), (2, "NORM/SILU_QA")):
        if which == 0 and variant >= 2:
            continue
        b = _GemvBench(); err = C.create_string_buffer(256)
        rc = L.lib().llamahip_bench_gemv(m._h, (variant << 4) | which, -1, 1, iters, C.byref(b), err, 256)
        row.append(f"{label} {b.ms_total * 1e3 / b.iters:6.2f} us" if rc == 0 else f"{label} failed: {err.value.decode()}")
    print(f"{name:9s} cold, back to back:  " + "   ".join(row))
m.close()

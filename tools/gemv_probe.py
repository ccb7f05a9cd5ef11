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
m.close()

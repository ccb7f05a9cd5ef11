#!/usr/bin/env python3
"""Prompt evaluation in fixed-size chunks (the reference feeds n_batch tokens per llama_eval): tokens/s
per chunk size, 7B synthetic model.  usage: chunk_probe.py [sizes...]   (default 2 4 8 9 16 32 64)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L
path = os.path.join(os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models"), "7B-seed20230312", "ggml-model-q4_0.bin")
sizes = [int(a) for a in sys.argv[1:]] or [2, 4, 8, 9, 16, 32, 64]
m = L.Model(path, n_ctx=512, flags=int(os.environ.get("FLAGS", "0")))
toks = np.random.default_rng(0).integers(3, 32000, 512).astype(np.int32); toks[0] = 1
for n in sizes:
    m.eval(toks[:n], 0)
    total = (480 // n) * n
    t0 = time.perf_counter()
    for c0 in range(0, total, n):
        m.eval(toks[c0:c0 + n], c0)
    dt = time.perf_counter() - t0
    print(f"chunk {n:3d}: {dt / (total // n) * 1e3:7.2f} ms per eval = {total / dt:7.0f} tok/s", flush=True)
m.close()

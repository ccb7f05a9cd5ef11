#!/usr/bin/env python3
"""The reference's prompt flow in isolation: 9-token evals at growing n_past (for rocprofv3 runs)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L
path = os.path.join(os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models"), "7B-seed20230312", "ggml-model-q4_0.bin")
m = L.Model(path, n_ctx=512, flags=int(os.environ.get("FLAGS", "0")))
toks = np.random.default_rng(0).integers(3, 32000, 512).astype(np.int32); toks[0] = 1
m.eval(toks[:9], 0)
t0 = time.perf_counter()
for c0 in range(0, 495, 9):
    m.eval(toks[c0:c0 + 9], c0)
dt = time.perf_counter() - t0
print(f"55 evals of 9 tokens: {dt / 55 * 1e3:.2f} ms per eval = {495 / dt:.0f} tok/s", flush=True)
m.close()

#!/usr/bin/env python3
"""One prompt eval of N tokens on the synthetic 7B model (for rocprofv3 runs). usage: prefill_one.py [N] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
path = os.path.join(os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models"), "7B-seed20230312", "ggml-model-q4_0.bin")
m = L.Model(path, n_ctx=max(512, N + 16))
toks = np.random.default_rng(0).integers(3, 32000, N).astype(np.int32); toks[0] = 1
m.eval(toks, 0)
t0 = time.perf_counter()
for _ in range(reps):
    m.eval(toks, 0)
dt = (time.perf_counter() - t0) / reps
print(f"eval of {N} tokens: {dt * 1e3:.1f} ms = {N / dt:.0f} tok/s")
m.close()

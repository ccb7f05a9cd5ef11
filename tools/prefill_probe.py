#!/usr/bin/env python3
"""Times prompt evaluation on the synthetic 7B model: the reference's 9-token chunks (n_ctx 512) and
one 2048-token eval (n_ctx 2560, BASELINE.json configs[2]).  Exact (bit-identical) path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L  # noqa: E402

path = os.path.join(os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models"), "7B-seed20230312", "ggml-model-q4_0.bin")
rng = np.random.default_rng(0)
m = L.Model(path, n_ctx=2560)
toks = rng.integers(3, 32000, 2048).astype(np.int32); toks[0] = 1
m.eval(toks[:9], 0)                                   # warm up
m.eval(toks[:64], 0)                                  # builds the prompt-only weight copies (first eval of 61+ rows)
t0 = time.perf_counter()
n_past = 0
for c0 in range(0, 504, 9):
    m.eval(toks[c0:c0 + 9], n_past); n_past += len(toks[c0:c0 + 9])
dt = time.perf_counter() - t0
print(f"prompt in 9-token chunks (reference behaviour): {n_past} tokens in {dt * 1e3:.1f} ms = {n_past / dt:.0f} tok/s")
m.eval(np.array([0, 1, 2, 3], np.int32), 0)
m.eval_chunks(toks[:504], 0, 9)
t0 = time.perf_counter(); m.eval_chunks(toks[:504], 0, 9); dt = time.perf_counter() - t0
print(f"the same 9-token chunks in one chunk-exact pass (llamahip_eval_chunks): 504 tokens in {dt * 1e3:.1f} ms = {504 / dt:.0f} tok/s")
for N in (64, 512, 2048):
    m.eval(toks[:N], 0)                               # (the first eval of a size allocates its attention workspace)
    t0 = time.perf_counter(); m.eval(toks[:N], 0); dt = time.perf_counter() - t0
    print(f"one eval of {N:5d} tokens: {dt * 1e3:8.1f} ms = {N / dt:7.0f} tok/s")
m.close()

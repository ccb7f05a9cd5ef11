#!/usr/bin/env python3
"""BASELINE.json configs[3]: mixed prefill + decode on one synthetic model (default LLaMA-13B), the run
that is captured under rocprofv3 for HBM GB/s.  Phases: a 495-token prompt in the reference's 9-token
chunks, one 512-token eval (n_past 0), then 256 greedy decode tokens.  Exact (bit-identical) path.
usage: mixed_run.py [7B|13B|30B|65B] [n_decode] [prompt_len]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (model table + synthetic file cache)
import llama_swift_amd as L  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "13B"
n_dec = int(sys.argv[2]) if len(sys.argv) > 2 else 256
P = int(sys.argv[3]) if len(sys.argv) > 3 else 512            # prompt length of the one-eval phase
cfg = bench.MODELS[name]
path = bench.model_path(name, cfg, 20230312)
t0 = time.perf_counter()
# rocprofv3 counter collection crashes on hipGraph launches: the --pmc passes run the same kernels eagerly
m = L.Model(path, n_ctx=max(1024, P + n_dec + 16), flags=1 if os.environ.get("LLAMAHIP_NO_GRAPH") else 0)
print(f"{name}: loaded in {time.perf_counter() - t0:.2f} s  {m.stats()}", flush=True)
rng = np.random.default_rng(4)
toks = rng.integers(3, cfg["n_vocab"], max(P, 512)).astype(np.int32); toks[0] = 1
m.eval(toks[:9], 0)
t0 = time.perf_counter(); n_past = 0
for c0 in range(0, 495, 9):
    m.eval(toks[c0:c0 + 9], n_past); n_past += 9
dt = time.perf_counter() - t0
print(f"prefill, 9-token chunks : {n_past} tokens in {dt * 1e3:8.1f} ms = {n_past / dt:8.1f} tok/s", flush=True)
t0 = time.perf_counter(); lg = m.eval(toks[:P], 0); dt = time.perf_counter() - t0
print(f"prefill, one {P} eval   : {P} tokens in {dt * 1e3:8.1f} ms = {P / dt:8.1f} tok/s   (FIRST eval of 61+ rows on this handle: it allocates and builds "
      f"the two prompt-only weight copies and the attention workspace)", flush=True)
t0 = time.perf_counter(); lg = m.eval(toks[:P], 0); dt = time.perf_counter() - t0
print(f"prefill, one {P} eval   : {P} tokens in {dt * 1e3:8.1f} ms = {P / dt:8.1f} tok/s   (again)", flush=True)
tok = int(np.argmax(lg))
w = m.decode_greedy(tok, P, 8)
t0 = time.perf_counter()
if os.environ.get("LLAMAHIP_NO_GRAPH"):      # counter passes: one synchronised step at a time (the profiler
    tk = int(w[-1])                           # segfaults with tens of thousands of dispatches in flight)
    for i in range(n_dec):
        tk = int(m.decode_greedy(tk, P + 8 + i, 1)[0])
else:
    out = m.decode_greedy(int(w[-1]), P + 8, n_dec)
dt = time.perf_counter() - t0
W = cfg["n_layer"] * (4 * cfg["n_embd"] ** 2 + 3 * cfg["n_embd"] * bench.n_ff(cfg)) * 20 // 32 + cfg["n_vocab"] * cfg["n_embd"] * 20 // 32
kv = cfg["n_layer"] * 2 * (P + 8 + n_dec / 2) * cfg["n_embd"] * 4
print(f"decode at context {P + 8}+  : {n_dec} tokens in {dt * 1e3:8.1f} ms = {n_dec / dt:8.1f} tok/s  "
      f"(algorithmic {((W + kv) * n_dec / dt) / 1e12:.2f} TB/s: weights {W / 1e9:.2f} GB + KV {kv / 1e9:.2f} GB per token)", flush=True)
m.close()

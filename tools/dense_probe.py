#!/usr/bin/env python3
"""Decode / prompt rate of an f16 model file with LLaMA-7B layer shapes (4 layers, so that the numpy writer
finishes in seconds): per-layer time extrapolates to the 32-layer model.  usage: dense_probe.py [n_layer]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
import llama_swift_amd as L
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
hp = synth.HParams(n_vocab=32000, n_embd=4096, n_mult=256, n_head=32, n_layer=nl)
path = "/tmp/dense7b_f16.bin"
rng = np.random.default_rng(1)
t = {}
for name, shape in synth.tensor_specs(hp):
    t[name] = (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)) if len(shape) == 1 else (0.02 * rng.standard_normal(shape, dtype=np.float32))
synth.write_model_unquantized(path, hp, t, 1)
del t
t0 = time.perf_counter(); m = L.Model(path, n_ctx=512); print(f"loaded {os.path.getsize(path) / 1e9:.2f} GB in {time.perf_counter() - t0:.2f} s")
prompt = np.concatenate([[1], rng.integers(3, 32000, 255)]).astype(np.int32)
m.eval(prompt[:8], 0)
t0 = time.perf_counter(); lg = m.eval(prompt, 0); dt = time.perf_counter() - t0
print(f"prompt 256 tokens: {dt * 1e3:.1f} ms ({nl} layers) -> {256 / (dt * 32 / nl):.0f} tok/s at 32 layers")
tok = int(np.argmax(lg)); m.decode_greedy(tok, 256, 4)
t0 = time.perf_counter(); m.decode_greedy(tok, 260, 64); dt = time.perf_counter() - t0
per_layer = dt / 64 / nl
wbytes = (4 * 4096 * 4096 + 3 * 4096 * 11008) * 2
print(f"decode: {dt / 64 * 1e3:.3f} ms/token for {nl} layers + lm head = {per_layer * 1e6:.1f} us/layer-ish; 32 layers ~ {1 / (dt / 64 * 32 / nl):.0f} tok/s; "
      f"layer weights {wbytes / 1e6:.0f} MB -> {wbytes / per_layer / 1e12:.2f} TB/s upper bound")
m.close()

#!/usr/bin/env python3
"""The in-process layer pipeline (ONE llamahip_model_load with a device list, include/llamahip.h) against the plain handle on ONE GPU:
every stage on device 0, so the figure shows what the stage launches, per-stage streams, events and hand-off copies cost with none of the
parallel hardware of a multi-GPU node (a single greedy stream is sequential through the stages anyway).  Tokens must be equal.
usage: inprocess_probe.py [model = 7B] [stages = 1,2,4,8] [tokens = 64]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import llama_swift_amd as L  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "7B"
stages = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
n_tok = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = bench.MODELS[model]
path = bench.model_path(model, cfg, 20230312)
rng = np.random.default_rng(3)
prompt = rng.integers(3, cfg["n_vocab"], 23).astype(np.int32)
prompt[0] = 1
ref = None
for S in stages:
    t0 = time.perf_counter()
    m = L.Model(path, n_ctx=512, devices=[0] * S if S > 1 else None, flags=int(os.environ.get("PROBE_FLAGS", "0")))      # PROBE_FLAGS=1: eager launches instead of captured graphs
    t_load = time.perf_counter() - t0
    m.eval(np.array([0, 1, 2, 3], np.int32), 0, 8)
    t0 = time.perf_counter()
    lg = m.eval_chunks(prompt, 0, 9, 8)
    t_prompt = time.perf_counter() - t0
    first = int(np.argmax(lg))
    m.decode_greedy(first, len(prompt), 4, 8)
    t0 = time.perf_counter()
    toks = m.decode_greedy(first, len(prompt), n_tok, 8)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    t = first
    for i in range(16):                                   # the bridge's loop: one llamahip_eval per token, logits to the host
        t = int(np.argmax(m.eval(np.array([t], np.int32), len(prompt) + i, 8)))
    dt_host = time.perf_counter() - t0
    if ref is None:
        ref = toks.tolist()
    print(f"{model} {S} stage(s) on device 0: load {t_load:6.1f} s | 23-token prompt in 9-token evals {t_prompt * 1e3:7.1f} ms | device-resident greedy loop "
          f"{n_tok / dt:7.1f} tokens/s ({dt / n_tok * 1e3:.3f} ms/token) | one llamahip_eval per token {16 / dt_host:7.1f} tokens/s | tokens equal to the first row: {toks.tolist() == ref}", flush=True)
    m.close()

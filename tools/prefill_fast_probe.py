#!/usr/bin/env python3
"""configs[2] (2048-token prompt in one eval, n_ctx 2560) on the exact path and with LLAMAHIP_FLAG_FAST_PREFILL
(measurement tooling): seconds, tokens/s, max |delta logit| of the last row, and where a 64-token greedy continuation
first leaves the exact path's.  usage: prefill_fast_probe.py [--model 7B]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import llama_swift_amd as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="7B")
ap.add_argument("--tokens", type=int, default=2048)
args = ap.parse_args()
cfg = bench.MODELS[args.model]
path = bench.model_path(args.model, cfg, 20230312)
rng = np.random.default_rng(9)
ptoks = rng.integers(3, cfg["n_vocab"], args.tokens).astype(np.int32)
ptoks[0] = 1
res = {}
for name, flags in (("exact", 0), ("fast", 16)):
    m = L.Model(path, n_ctx=args.tokens + 512, flags=flags)
    m.eval(ptoks, 0, 8)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); lg = m.eval(ptoks, 0, 8); best = min(best, time.perf_counter() - t0)
    toks = m.decode_greedy(int(np.argmax(lg)), args.tokens, 64, 8)
    res[name] = (best, lg, toks)
    m.close()
    print(f"{name:5s}: {args.tokens} tokens in {best * 1e3:7.1f} ms = {args.tokens / best:8.0f} tokens/s", flush=True)
e, f = res["exact"], res["fast"]
d = np.abs(e[1] - f[1])
bad = np.flatnonzero(e[2] != f[2])
top2 = np.partition(e[1], -2)[-2:]
print(f"fast vs exact, last-row logits: max |delta| {d.max():.3e}, mean |delta| {d.mean():.3e}, argmax equal: {int(np.argmax(e[1])) == int(np.argmax(f[1]))} "
      f"(exact top-2 margin {top2[1] - top2[0]:.3e}); 64-token greedy continuation: "
      + ("identical" if not bad.size else f"first divergence at token {int(bad[0])}") + f"; speed-up {e[0] / f[0]:.2f}x")

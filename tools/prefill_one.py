#!/usr/bin/env python3
"""One long prompt eval on the synthetic 7B (measurement tooling): tokens/s of a single eval of N tokens, exact path.
usage: prefill_one.py [N=2048] [reps=3]     (LLAMAHIP_MFMA_I8=1: the int8 matrix-core kernel of round 1)"""
import os
import sys
import time

if not os.environ.get("LLAMAHIP_WITH_TORCH"):      # (rocprofv3 crashes on the system HIP runtime here: profile with torch's)
    os.environ.setdefault("LLAMAHIP_NO_TORCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import llama_swift_amd as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = bench.MODELS["7B"]
path = bench.model_path("7B", cfg, 20230312)
m = L.Model(path, n_ctx=N + 512)
rng = np.random.default_rng(9)
p = rng.integers(3, cfg["n_vocab"], N).astype(np.int32)
p[0] = 1
lg = m.eval(p, 0, 8)
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter(); lg = m.eval(p, 0, 8); best = min(best, time.perf_counter() - t0)
import zlib
print(f"{N} tokens in {best * 1e3:7.1f} ms = {N / best:8.0f} tokens/s   logits crc {zlib.crc32(lg.tobytes()):08x}   {'int8 MFMA kernel' if os.environ.get('LLAMAHIP_MFMA_I8') else 'fp16 two-chain MFMA kernel'}")
m.close()

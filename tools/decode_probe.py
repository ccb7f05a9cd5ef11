#!/usr/bin/env python3
"""Decode-step probe (measurement tooling): loads the synthetic model once and times greedy decode on the
device at one or more context offsets.  Variants of the library are selected through the environment
(LLAMAHIP_LIB, LLAMAHIP_* switches), one process per variant -- see tools/decode_ab.sh.
usage: decode_probe.py [--model 7B] [--steps 64] [--at 8,256] [--reps 3] [--check]
  --check   also prints a CRC of the generated tokens (variants must agree)"""
import argparse
import os
import sys
import time
import zlib

if not os.environ.get("LLAMAHIP_WITH_TORCH"):      # (rocprofv3 crashes on the system HIP runtime here: profile with torch's)
    os.environ.setdefault("LLAMAHIP_NO_TORCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import llama_swift_amd as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="7B")
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--at", default="8")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--n_ctx", type=int, default=512)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--flags", type=int, default=int(os.environ.get("PROBE_FLAGS", "0")), help="LLAMAHIP_FLAG_* (1 = eager launches instead of hipGraph replay)")
args = ap.parse_args()
cfg = bench.MODELS[args.model]
path = bench.model_path(args.model, cfg, 20230312)
m = L.Model(path, n_ctx=args.n_ctx, flags=args.flags)
rng = np.random.default_rng(11)
out = []
for at in [int(x) for x in args.at.split(",")]:
    prompt = rng.integers(3, cfg["n_vocab"], at).astype(np.int32)
    prompt[0] = 1
    logits = m.eval(prompt, 0, args.threads)
    tok = int(np.argmax(logits))
    steps = min(args.steps, args.n_ctx - at)
    m.decode_greedy(tok, at, min(4, steps), args.threads)          # graph instantiation + warm-up
    best, toks = 1e9, None
    for _ in range(args.reps):
        t0 = time.perf_counter()
        toks = m.decode_greedy(tok, at, steps, args.threads)
        best = min(best, time.perf_counter() - t0)
    out.append(f"ctx {at}..{at + steps}: {steps / best:7.1f} tok/s {best / steps * 1e3:6.3f} ms/tok crc {zlib.crc32(np.asarray(toks, np.int32).tobytes()):08x}")
print(" | ".join(out), f"| lut_math {L.lib().llamahip_debug_lut_math()}", flush=True)
m.close()

#!/usr/bin/env python3
"""Batched decode steps in isolation (measurement tooling): S sequences at different positions stepped as ONE set per step
(llamahip_stage_step_set) on one handle -- ms per step, aggregate tokens/s, a CRC of every sequence's tokens (variants of the
library / of its switches must agree), and with --evals the reference's 9-token evals next to it.  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel table of a set step (tools/set_ab.sh).
usage: set_probe.py [--model 7B] [--seqs 4,8] [--steps 96] [--n_ctx 512] [--evals 9]"""
import argparse
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import llama_swift_amd as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="7B")
ap.add_argument("--seqs", default="4")
ap.add_argument("--steps", type=int, default=96)
ap.add_argument("--n_ctx", type=int, default=512)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--layers", type=int, default=0, help="load only the first N layers (a pipeline stage's share)")
ap.add_argument("--evals", default="", help="also time llama_eval calls of these row counts (e.g. 9: the reference's prompt flow)")
args = ap.parse_args()
cfg = bench.MODELS[args.model]
path = bench.model_path(args.model, cfg, 20230312)
rng = np.random.default_rng(5)
kw = dict(layer_begin=0, layer_end=args.layers) if args.layers else {}
for S in [int(x) for x in args.seqs.split(",") if x]:
    m = L.Model(path, n_ctx=args.n_ctx, n_seq=S, **kw)
    last = not args.layers or args.layers >= cfg["n_layer"]
    prompts = [np.concatenate([[1], rng.integers(3, cfg["n_vocab"], 7 + s)]).astype(np.int32) for s in range(S)]      # positions 8, 9, ...
    firsts = []
    hid = [torch.zeros(cfg["n_embd"], dtype=torch.float32, device="cuda") for _ in range(S)]
    for s in range(S):
        m.set_seq(s)
        if last:
            firsts.append(int(np.argmax(m.eval(prompts[s], 0, args.threads))))
        else:
            h = torch.zeros(len(prompts[s]) * cfg["n_embd"], dtype=torch.float32, device="cuda")
            m.eval_stage(0, tokens=prompts[s], hidden_out=h.data_ptr(), n_threads=args.threads)
            firsts.append(5 + s)
    m.set_seq(0)
    bufs = [torch.tensor([firsts[s]], dtype=torch.int32, device="cuda") for s in range(S)]
    st = torch.cuda.current_stream().cuda_stream
    seqs = list(range(S))

    def bind():
        for s in range(S):
            bufs[s].fill_(firsts[s])
            if last:
                m.stage_bind(s, len(prompts[s]), token_in=bufs[s].data_ptr(), token_out=bufs[s].data_ptr())
            else:
                m.stage_bind(s, len(prompts[s]), token_in=bufs[s].data_ptr(), hidden_out=hid[s].data_ptr())
    steps = min(args.steps, args.n_ctx - (8 + S) - 8)
    bind()
    for _ in range(4):
        m.stage_step_set(seqs, args.threads, st)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        bind()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.stage_step_set(seqs, args.threads, st)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    crc = 0
    if last:
        for s in range(S):
            n, pos, got = m.stage_trace(s, steps)
            crc = zlib.crc32(np.asarray(got, np.int32).tobytes(), crc)
    print(f"set of {S:2d}: {best / steps * 1e3:6.3f} ms per step = {S * steps / best:7.1f} tok/s aggregate, crc {crc:08x}, paths {L.gemm_paths()}", flush=True)
    m.close()
if args.evals:
    m = L.Model(path, n_ctx=args.n_ctx)
    toks = np.random.default_rng(0).integers(3, cfg["n_vocab"], args.n_ctx).astype(np.int32)
    toks[0] = 1
    for n in [int(x) for x in args.evals.split(",")]:
        lg = m.eval(toks[:n], 0, args.threads)
        total = ((args.n_ctx - 32) // n) * n
        t0 = time.perf_counter()
        for c0 in range(0, total, n):
            lg = m.eval(toks[c0:c0 + n], c0, args.threads)
        dt = time.perf_counter() - t0
        print(f"evals of {n:2d} tokens: {dt / (total // n) * 1e3:6.3f} ms per eval = {total / dt:7.0f} tok/s, crc {zlib.crc32(lg.tobytes()):08x}", flush=True)
    m.close()

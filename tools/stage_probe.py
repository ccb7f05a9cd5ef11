#!/usr/bin/env python3
"""Per-stage step time of the layer pipeline on ONE GPU: loads the layer range a rank of an N-stage
pipeline would own, binds S sequences and times stream-ordered steps (no RCCL: hand-off buffers are
left as they are).  The slowest stage bounds the N-GPU aggregate rate: tokens/s <= 1 / t_stage.
usage: stage_probe.py [model-file] [n_layer] [world ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from llama_swift_amd import binding as L
from llama_swift_amd.pipeline import layer_range

path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else os.path.join(os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models"), "7B-seed20230312", "ggml-model-q4_0.bin")
n_layer = int(sys.argv[2]) if len(sys.argv) > 2 else 32
worlds = [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8]
for world in worlds:
    S = world
    worst = 0.0
    for rank in sorted({0, world // 2, world - 1}):
        lo, hi = layer_range(n_layer, rank, world)
        m = L.Model(path, n_ctx=512, device=0, layer_begin=lo, layer_end=hi, n_seq=S)
        first, last = lo == 0, hi == n_layer
        tok = [torch.full((1,), 5 + s, dtype=torch.int32, device="cuda") for s in range(S)]
        hin = [torch.randn(m.n_embd, device="cuda") for s in range(S)]
        hout = [torch.zeros(m.n_embd, device="cuda") for s in range(S)]
        torch.cuda.synchronize()
        for s in range(S):
            m.stage_bind(s, 16, token_in=tok[s].data_ptr() if first else 0, hidden_in=0 if first else hin[s].data_ptr(),
                         hidden_out=0 if last else hout[s].data_ptr(), token_out=tok[s].data_ptr() if last else 0)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            for s in range(S):
                m.stage_step(s, 8, st)
        torch.cuda.synchronize()
        rounds = 200
        t0 = time.perf_counter()
        for _ in range(rounds):
            for s in range(S):
                m.stage_step(s, 8, st)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        us = dt / (rounds * S) * 1e6
        worst = max(worst, us)
        print(f"world {world} rank {rank} layers [{lo},{hi}): {us:7.1f} us/step (host enqueue {t_host / (rounds * S) * 1e6:5.1f} us/step), positions 16..{16 + rounds + 3}", flush=True)
        m.close()
    print(f"world {world}: slowest stage {worst:.1f} us -> compute-bound aggregate <= {1e6 / worst:.0f} tokens/s", flush=True)

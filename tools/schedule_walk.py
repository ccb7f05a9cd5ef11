#!/usr/bin/env python3
"""Every launch of the DEFAULT schedules (no environment switch) at the four LLaMA widths, for tools/kernel_scratch_report.py: 2-layer
models of the 7B / 13B / 30B / 65B widths run through each entry point a caller reaches -- the bridge's prompt flow (4-token warm-up, 9-token
evals, llamahip_eval_chunks), a 70- and a 600-token eval (matrix-core kernels, lane-per-query attention), single-token evals and the
device-resident greedy loop across every position threshold of the decode attention schedule, set steps of 4 and 8 sequences, and the
same through a 2-stage in-process pipeline handle (stage launches: the first layer of a later stage takes its row without the
producer's partial sums).  Run under `rocprofv3 --kernel-trace --stats`; the kernel names of the stats table are the answer.
usage: schedule_walk.py [widths = 7B,13B,30B,65B]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import llama_swift_amd as L  # noqa: E402
import synth  # noqa: E402
from conftest import synth_tool  # noqa: E402

WIDTHS = {"7B": (4096, 32), "13B": (5120, 40), "30B": (6656, 52), "65B": (8192, 64)}
want = (sys.argv[1] if len(sys.argv) > 1 else "7B,13B,30B,65B").split(",")
import torch  # noqa: E402  (device buffers of the set steps)

with tempfile.TemporaryDirectory() as td:
    for name in want:
        d, H = WIDTHS[name]
        V, n_ctx = 32000, 2304
        path = synth_tool(os.path.join(td, f"{name}.bin"), seed=5, n_vocab=V, n_embd=d, n_mult=256, n_head=H, n_layer=2)
        for devices in (None, [0, 0]):
            with L.Model(path, n_ctx=n_ctx, devices=devices, n_seq=8 if devices is None else 1) as m:
                m.eval(np.array([0, 1, 2, 3], np.int32), 0, 8)
                p = synth.synth_prompt(700, V, seed=1)
                for c0 in range(0, 27, 9):
                    m.eval(p[c0:c0 + 9], c0, 8)
                m.eval_chunks(p[:600], 0, 9, 8)
                m.eval(p[:70], 0, 8)
                lg = m.eval(p[:600], 0, 8)
                t, pos = int(np.argmax(lg)), 600
                # decode across every default threshold of attn_sched_at (13B: 544, 1600; 65B: 448, 2048; 7B: 1280): host-driven and device loop
                for start in (8, 440, 540, 1275, 1596, 2044):
                    m.eval(p[:1], 0, 8)
                    lg = m.eval(np.array([t], np.int32), start, 8)
                    m.decode_greedy(int(np.argmax(lg)), start + 1, 12, 8)
                if devices is None:
                    bufs = [torch.tensor([5 + s], dtype=torch.int32, device="cuda") for s in range(8)]
                    for s in range(8):
                        m.set_seq(s)
                        m.eval(p[:3 + s], 0, 8)
                        m.stage_bind(s, 3 + s, token_in=bufs[s].data_ptr(), token_out=bufs[s].data_ptr())
                    m.set_seq(0)
                    st = torch.cuda.current_stream().cuda_stream
                    for S in (2, 4, 8):
                        for _ in range(3):
                            m.stage_step_set(list(range(S)), 8, st)
                    m.stage_step(0, 8, st)
                    torch.cuda.synchronize()
        print("walked", name, flush=True)

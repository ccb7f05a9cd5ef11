#!/usr/bin/env python3
"""How long a pipeline stage waits for a mailbox row / token that never comes, with the PRODUCTION poll bounds (the bench's one-token
handshake pays this once before it falls back to RCCL when a peer mapping does not carry the stores): the consumer stage steps without
its producer; the first stage steps twice without token feedback.  usage: tools/mailbox_silence_probe.py [model file to write]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
import llama_swift_amd as L
hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=4)
p = sys.argv[1] if len(sys.argv) > 1 else "/tmp/silent_m.bin"
synth.write_model(p, hp, synth.random_tensors(hp, seed=9))
a = L.Model(p, n_ctx=64, layer_begin=0, layer_end=1); b = L.Model(p, n_ctx=64, layer_begin=1, layer_end=4)
_, a_tok, _, _ = a.stage_mailbox(0); b_hid, _, _, _ = b.stage_mailbox(0)
a.stage_mailbox_connect(0, next_hidden_ptr=b_hid); b.stage_mailbox_connect(0, token_ptr=a_tok)
tok = torch.ones(1, dtype=torch.int32, device='cuda'); torch.cuda.synchronize()
a.stage_bind(0, 0, token_in=tok.data_ptr()); b.stage_bind(0, 0)
t0 = time.time()
b.stage_step(0, 8, 0)            # the producer never steps
try:
    b.stage_trace(0, 1); print('NO ERROR')
except L.LlamaHipError as e:
    print('ERR', e.code, str(e)[:120]); print('SECONDS consumer without producer', time.time() - t0)
# the first stage waiting for a token that never comes back (second token of the sequence)
t0 = time.time()
a.stage_bind(0, 0, token_in=tok.data_ptr())
a.stage_step(0, 8, 0); a.stage_step(0, 8, 0)
try:
    a.stage_trace(0, 1); print('NO ERROR (first stage has no trace of its own)')
except L.LlamaHipError as e:
    print('ERR', e.code, str(e)[:120])
print('SECONDS first stage without token feedback', time.time() - t0)

#!/usr/bin/env python3
"""Golden vectors for Q4_1 model files (f16 = 3) from the REFERENCE ITSELF: the f16 input is quantized by
the reference's quantize tool (oracle/_ref/quantize <in> <out> 3) and evaluated by the reference's ggml.c
(oracle/_ref/libggml_ref.so through oracle/ref_driver.cpp).  Build container only.  Stored: the f16 input
file, the Q4_1 file the reference wrote, prompt, all-row logits, greedy tokens, one layer's KV rows."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import reflib  # noqa: E402
import synth  # noqa: E402

R = reflib.RefLib()
REFQ = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "quantize")
hp = synth.HParams(n_vocab=32, n_embd=128, n_mult=64, n_head=1, n_layer=2)
t = synth.random_tensors(hp, seed=4141)
w = t["layers.0.attention.wq.weight"]
w[0, :32] = 0.0                                         # d = 0 block
w[1, :32] = -np.arange(1, 33, dtype=np.float32)         # all negative: the offline quantizer's max starts at FLT_MIN
w[2, :32] = np.arange(32, dtype=np.float32) / 2.0 + 0.25
prompt = synth.synth_prompt(33, hp.n_vocab, seed=7)
out = {"prompt": prompt}
with tempfile.TemporaryDirectory() as td:
    src, dst = os.path.join(td, "in.bin"), os.path.join(td, "q41.bin")
    synth.write_model_unquantized(src, hp, t, 1)
    subprocess.run([REFQ, src, dst, "3"], check=True, stdout=subprocess.DEVNULL)
    out["in_f16"] = np.fromfile(src, np.uint8)
    out["q41_file"] = np.fromfile(dst, np.uint8)
    for nth in (8, 3):
        m = R.load(dst, 64)
        r = m.eval(prompt[:20], 0, nth, all_logits=True)
        out[f"nth{nth}_logits_a"] = r["logits_all"]
        r = m.eval(prompt[20:], 20, nth, all_logits=True)
        out[f"nth{nth}_logits_b"] = r["logits_all"]
        tok, toks, n_past = int(np.argmax(r["logits"])), [], len(prompt)
        for i in range(10):
            toks.append(tok)
            lo = m.eval(np.array([tok], np.int32), n_past + i, nth)["logits"]
            tok = int(np.argmax(lo))
        out[f"nth{nth}_tokens"] = np.array(toks + [tok], np.int32)
        out[f"nth{nth}_logits_last"] = lo
        k, v = m.kv(1, n_past + 10)
        out[f"nth{nth}_k1"], out[f"nth{nth}_v1"] = k, v
        m.close()
np.savez_compressed(os.path.join(HERE, "q41_model.npz"), **out)
print("q41_model.npz:", os.path.getsize(os.path.join(HERE, "q41_model.npz")) // 1024, "KiB")

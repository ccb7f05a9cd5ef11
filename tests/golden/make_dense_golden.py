#!/usr/bin/env python3
"""Golden vectors for f16 / f32 model files from the REFERENCE ITSELF
(oracle/_ref/libggml_ref.so: the reference's ggml.c running the reference graph, oracle/ref_driver.cpp).
Build container only.  The model is regenerated from its seed by tests/synth.py on both sides; only the
prompt, the expected logits / KV rows / greedy tokens are stored."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import reflib  # noqa: E402
import synth  # noqa: E402

R = reflib.RefLib()
hp = synth.HParams(n_vocab=96, n_embd=256, n_mult=64, n_head=2, n_layer=2)
t = synth.random_tensors(hp, seed=606)
prompt = synth.synth_prompt(37, hp.n_vocab, seed=11)
out = {"prompt": prompt, "hp": np.array([hp.n_vocab, hp.n_embd, hp.n_mult, hp.n_head, hp.n_layer], np.int32), "seed": np.array([606])}
with tempfile.TemporaryDirectory() as td:
    for tag, ftype in (("f16", 1), ("f32", 0)):
        path = os.path.join(td, f"m_{tag}.bin")
        synth.write_model_unquantized(path, hp, t, ftype)
        for nth in (8, 3):
            m = R.load(path, 64)
            r = m.eval(prompt[:28], 0, nth, all_logits=True)
            out[f"{tag}_nth{nth}_logits_a"] = r["logits_all"]
            r = m.eval(prompt[28:], 28, nth, all_logits=True)           # continuation at n_past > 0
            out[f"{tag}_nth{nth}_logits_b"] = r["logits_all"]
            tok, toks, n_past = int(np.argmax(r["logits"])), [], len(prompt)
            for i in range(12):
                toks.append(tok)
                lo = m.eval(np.array([tok], np.int32), n_past + i, nth)["logits"]
                tok = int(np.argmax(lo))
            out[f"{tag}_nth{nth}_tokens"] = np.array(toks + [tok], np.int32)
            out[f"{tag}_nth{nth}_logits_last"] = lo
            k, v = m.kv(1, n_past + 12)
            out[f"{tag}_nth{nth}_k1"], out[f"{tag}_nth{nth}_v1"] = k, v
            m.close()
np.savez_compressed(os.path.join(HERE, "dense_model.npz"), **out)
print("dense_model.npz:", os.path.getsize(os.path.join(HERE, "dense_model.npz")) // 1024, "KiB")

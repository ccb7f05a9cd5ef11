#!/usr/bin/env python3
"""Golden files for llamahip_quantize_file, produced by the REFERENCE'S OWN quantize
tool: Sources/cpp/quantize.cpp + utils.cpp + ggml.c compiled in place (oracle/_ref/quantize, recipe in
oracle/Makefile).  Build container only; the .npz (input model files + the files the reference wrote)
is data and is committed.

Inputs: one tiny synthetic model as an f16 file and as an f32 file (tests/synth.py writer, the format
tools/convert-pth-to-ggml.py emits).  Some rows carry crafted values: blocks whose scaled elements land
exactly on .5 (round-half-away vs round-half-even differ there), all-zero blocks, a block with a
huge outlier, denormal-small blocks.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

REFQ = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "quantize")
hp = synth.HParams(n_vocab=32, n_embd=64, n_mult=32, n_head=1, n_layer=1)
t = synth.random_tensors(hp, seed=4242)
w = t["layers.0.attention.wq.weight"]
w[0, :32] = 0.0                                                        # d = 0 -> id = 0
w[1, :32] = np.array([7.0] + [0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 3.5, -3.5] * 3 + [6.5] * 7, np.float32)   # amax 7: id = 1, exact .5 ties
w[2, :32] = np.array([1e4] + [1.0] * 31, np.float32)                   # outlier squeezes the rest to 0
w[3, :32] = 1e-7 * np.arange(32, dtype=np.float32)                     # tiny (f16: denormal / flushed)
w[4, :32] = -np.arange(32, dtype=np.float32)                           # negative amax side
out = {}
with tempfile.TemporaryDirectory() as td:
    for tag, ftype in (("f16", 1), ("f32", 0)):
        src, dst = os.path.join(td, f"in_{tag}.bin"), os.path.join(td, f"out_{tag}.bin")
        synth.write_model_unquantized(src, hp, t, ftype)
        subprocess.run([REFQ, src, dst, "2"], check=True, stdout=subprocess.DEVNULL)
        out[f"in_{tag}"] = np.fromfile(src, np.uint8)
        out[f"out_{tag}"] = np.fromfile(dst, np.uint8)
        print(tag, out[f"in_{tag}"].size, "->", out[f"out_{tag}"].size, "bytes")
np.savez_compressed(os.path.join(HERE, "quantize_file.npz"), **out)
print("quantize_file.npz:", os.path.getsize(os.path.join(HERE, "quantize_file.npz")) // 1024, "KiB")

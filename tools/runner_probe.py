#!/usr/bin/env python3
"""LlamaRunner.run end to end (the reference's user-facing flow: 8-token prompt batches, one eval + one top-k / top-p
sample per token, token text through the callback) -- generated tokens per second and a CRC of the text, with the
sampler's candidate selection on the device (default) or on the host (LLAMAHIP_HOST_SAMPLER=1).  Measurement tooling."""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from llama_swift_amd import Config, LlamaRunner  # noqa: E402

cfg = bench.MODELS["7B"]
path = bench.model_path("7B", cfg, 20230312)
rng = np.random.default_rng(7)
stamps, toks = [], []
rn = LlamaRunner(path)
c = Config(numThreads=8, numTokens=320, n_ctx=512, keepModel=True)
text = "".join("tok%05d" % t for t in rng.integers(3, cfg["n_vocab"], 16))
rn.run(text, c)
rn.run(text, c, tokenHandler=lambda t: (stamps.append(time.perf_counter()), toks.append(t if isinstance(t, bytes) else str(t).encode())))
rn.close()
gen = stamps[-257:]
print(("host sampler  " if os.environ.get("LLAMAHIP_HOST_SAMPLER") else "device top-k  ") + f"{256 / (gen[-1] - gen[0]):7.1f} sampled tokens/s   text crc {zlib.crc32(b''.join(toks)):08x}")

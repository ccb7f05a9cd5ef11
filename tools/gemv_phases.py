#!/usr/bin/env python3
"""Where the time goes INSIDE the decode GEMV kernels, measured in the real decode loop (graph
replay): s_memtime stamps of one thread of the middle workgroup of every k_gemv launch."""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L  # noqa: E402

path = os.path.join(os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models"), "7B-seed20230312", "ggml-model-q4_0.bin")
m = L.Model(path, n_ctx=512)
n_past = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prompt = (np.arange(n_past, dtype=np.int32) * 7919 + 13) % 32000
prompt[0] = 1
for c0 in range(0, n_past, 64):
    lg = m.eval(prompt[c0:c0 + 64], c0)
lib = L.lib()
lib.llamahip_debug_decode_phases.restype = C.c_int64
lib.llamahip_debug_decode_phases.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t]
cap = 4096
rec = np.zeros((cap, 8), np.uint64)
err = C.create_string_buffer(512)
n = lib.llamahip_debug_decode_phases(m._h, n_past, int(np.argmax(lg)), 8, rec.ctypes.data_as(C.c_void_p), cap, err, 512)
assert n > 0, err.value
rec = rec[:n].astype(np.int64)
groups = collections.defaultdict(list)
for r in rec:
    groups[(int(r[5]), int(r[6]), int(r[7]))].append(np.diff(r[:5]))
pre = {0: "QA", 1: "PLAIN", 2: "NORM", 3: "SILU_MUL"}
epi = {0: "STORE", 1: "RESID", 2: "SILU_QA"}
print(f"{n} k_gemv launches probed at n_past={n_past}; cycles (s_memtime), median over launches")
for (ng, nc, pe), v in sorted(groups.items()):
    d = np.median(np.array(v), axis=0)
    print(f"M={ng * 8:6d} K={nc * 256:6d} {pre[pe >> 4]:5s}/{epi[pe & 15]:8s} n={len(v):4d}: entry->issued {d[0]:6.0f} | prologue {d[1]:6.0f} | consume {d[2]:6.0f} | epilogue {d[3]:6.0f} | total {d.sum():6.0f}")
m.close()

#!/usr/bin/env python3
"""Timeline of the decode mat-vec launches in the real decode loop (graph replay): EVERY workgroup of every
k_gemv launch records s_memtime at entry / loads issued / prologue done / weights consumed / exit
(libllamahip_probe3.so, `make probe`).  Records are split into launches by time (launches of one stream are
serialised) and summarised per kernel kind: how long the grid takes to start, how the phases of a median
workgroup line up, and how ragged the end is.
usage: LLAMAHIP_LIB=libllamahip_probe3.so tools/gemv_timeline.py [n_past] [steps]"""
import collections
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("LLAMAHIP_LIB", "libllamahip_probe3.so")
os.environ.setdefault("LLAMAHIP_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L  # noqa: E402

path = os.path.join(os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models"), "7B-seed20230312", "ggml-model-q4_0.bin")
m = L.Model(path, n_ctx=512)
n_past = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prompt = (np.arange(n_past, dtype=np.int32) * 7919 + 13) % 32000
prompt[0] = 1
for c0 in range(0, n_past, 64):
    lg = m.eval(prompt[c0:c0 + 64], c0)
lib = L.lib()
lib.llamahip_debug_decode_phases.restype = C.c_int64
lib.llamahip_debug_decode_phases.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t]
cap = 60000 * steps
rec = np.zeros((cap, 8), np.uint64)
err = C.create_string_buffer(512)
n = lib.llamahip_debug_decode_phases(m._h, n_past, int(np.argmax(lg)), steps, rec.ctypes.data_as(C.c_void_p), cap, err, 512)
assert n > 0, err.value
rec = rec[:n].astype(np.int64)
rec = rec[np.argsort(rec[:, 7], kind="stable")]
# s_memtime (per-XCD counters) -> microseconds through the 100 MHz wall clock sampled at entry and exit
tpu = float((rec[:, 4] - rec[:, 0]).sum()) / (float((rec[:, 6] - rec[:, 7]).sum()) / 100.0)
kind = rec[:, 5] >> 48
grp = (rec[:, 5] >> 32) & 0xffff
w0 = rec[:, 7] / 100.0                          # entry, us
w1 = rec[:, 6] / 100.0                          # exit, us
launches = []
i = 0
while i < n:
    j = i + 1
    end = w1[i]
    while j < n and kind[j] == kind[i] and grp[j] == grp[i] and w0[j] < end + 0.2:       # a later launch cannot start before this one ended
        end = max(end, w1[j]); j += 1
    launches.append((i, j))
    i = j
pre = {0: "QA", 1: "PLAIN", 2: "NORM", 3: "SILU_MUL", 4: "NORMP"}
epi = {0: "STORE", 1: "RESID", 2: "SILU_QA"}
agg = collections.defaultdict(list)
prev_end = None
for (a, b) in launches:
    r = rec[a:b]
    t0 = w0[a:b].min()
    row = dict(nwg=b - a, gap=(t0 - prev_end) if prev_end is not None else 0.0,
               start_spread=w0[a:b].max() - t0,
               issued=np.median(r[:, 1] - r[:, 0]) / tpu, prologue=np.median(r[:, 2] - r[:, 1]) / tpu,
               consume=np.median(r[:, 3] - r[:, 2]) / tpu, epilogue=np.median(r[:, 4] - r[:, 3]) / tpu,
               wg_total=np.median(r[:, 4] - r[:, 0]) / tpu, first_end=w1[a:b].min() - t0,
               p50_end=np.median(w1[a:b]) - t0, p90_end=np.percentile(w1[a:b], 90) - t0, span=w1[a:b].max() - t0)
    prev_end = w1[a:b].max()
    agg[(int(kind[a]), 0, int(grp[a]))].append(row)
print(f"{n} workgroup records, {len(launches)} launches at n_past={n_past}; s_memtime = {tpu:.1f} ticks/us; medians over launches, microseconds")
print(f"{'kernel':28s} {'launches':>8s} {'WGs':>5s} {'gap<-':>6s} {'ramp':>6s} | {'issue':>6s} {'prolog':>6s} {'consum':>6s} {'epilog':>6s} {'WG':>6s} | {'1st end':>7s} {'p50 end':>7s} {'p90 end':>7s} {'span':>6s}")
for (k, ng, nc), rows in sorted(agg.items()):
    med = lambda f: float(np.median([r[f] for r in rows]))
    name = f"K={nc * 256} {pre.get(k >> 4, '?')}/{epi.get(k & 15, '?')}"
    print(f"{name:28s} {len(rows):8d} {int(med('nwg')):5d} {med('gap'):6.2f} {med('start_spread'):6.2f} | {med('issued'):6.2f} {med('prologue'):6.2f} {med('consume'):6.2f} {med('epilogue'):6.2f} {med('wg_total'):6.2f} | "
          f"{med('first_end'):7.2f} {med('p50_end'):7.2f} {med('p90_end'):7.2f} {med('span'):6.2f}")
m.close()

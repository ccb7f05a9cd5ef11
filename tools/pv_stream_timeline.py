#!/usr/bin/env python3
"""Timeline of the long-context decode attention (k_dec_scores + k_dec_pv_stream) in the real decode loop: every workgroup records
s_memtime at its phase boundaries (libllamahip_probe3.so, `make probe`); launches are lined up on the 100 MHz wall clock.
  k_dec_scores    : entry | q rotated, barrier passed | dots done
  k_dec_pv_stream : entry | score row arrived + max known | soft_max done | stage 0 in LDS | consumed | stage 1 in LDS | consumed |
                    stage 2 in LDS | all chains done | exit
usage: LLAMAHIP_ATTN_LONG_FROM=0 tools/pv_stream_timeline.py [n_past] [steps] [n_ctx]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("LLAMAHIP_LIB", "libllamahip_probe3.so")
os.environ.setdefault("LLAMAHIP_NO_TORCH", "1")
os.environ.setdefault("LLAMAHIP_ATTN_LONG_FROM", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L  # noqa: E402

import bench  # noqa: E402
path = bench.model_path("7B", bench.MODELS["7B"], 20230312)          # (writes the synthetic model file if it is not there yet)
n_past = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2560
m = L.Model(path, n_ctx=n_ctx)
prompt = (np.arange(n_past, dtype=np.int64) * 7919 + 13) % 32000
prompt = prompt.astype(np.int32); prompt[0] = 1
lg = m.eval(prompt, 0)
lib = L.lib()
lib.llamahip_debug_decode_phases.restype = C.c_int64
lib.llamahip_debug_decode_phases.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t]
cap = 200000 * steps
rec = np.zeros((cap, 8), np.uint64)
err = C.create_string_buffer(512)
n = lib.llamahip_debug_decode_phases(m._h, n_past, int(np.argmax(lg)), steps, rec.ctypes.data_as(C.c_void_p), cap, err, 512)
assert n > 0, err.value
rec = rec[:n].astype(np.int64)
kind = rec[:, 5] >> 48
sel = (kind == 0xB0) | (kind == 0xB1) | (kind == 0xB2)
rec, kind = rec[sel], kind[sel]
a = rec[kind == 0xB0]; b = rec[kind == 0xB1]; s = rec[kind == 0xB2]
print(f"records: scores {len(s)}, pv_stream {len(a)} (+{len(b)} second halves) at n_past={n_past}")
# s_memtime ticks per microsecond from the workgroups' own (entry stamp, wall clock) pairs
def tpu_of(r_first, r_last_stamp):
    return float((r_last_stamp - r_first[:, 0]).sum()) / (float((r_first[:, 6] - r_first[:, 7]).sum()) / 100.0)
key = lambda r: r[:, 7] * 4096 + (r[:, 5] & 0xffffffff) * 8 + ((r[:, 5] >> 32) & 7)
ia, ib = np.argsort(key(a), kind="stable"), np.argsort(key(b), kind="stable")
a, b = a[ia], b[ib]
assert len(a) == len(b) and np.all(a[:, 7] == b[:, 7])
tpu = tpu_of(a, b[:, 4])
us = lambda x: x / tpu
print(f"s_memtime = {tpu:.1f} ticks/us")
names = ["row arrived, max known", "soft_max done", "stage 0 in LDS", "stage 0 consumed", "stage 1 in LDS", "stage 1 consumed", "stage 2 in LDS", "all chains done", "exit"]
st = np.concatenate([a[:, 1:5], b[:, 0:5]], axis=1) - a[:, 0:1]
z = (a[:, 5] >> 32) & 7
for zz in sorted(set(z.tolist())):
    r = st[z == zz]
    print(f"k_dec_pv_stream workgroups z={zz} ({len(r)}): medians on the workgroup's own clock, us after entry")
    for j, nm in enumerate(names):
        col = r[:, j]; col = col[col > 0]
        if len(col): print(f"   {nm:26s} {us(np.median(col)):7.2f}   (p10 {us(np.percentile(col, 10)):6.2f}, p90 {us(np.percentile(col, 90)):6.2f})")
# launches: group by wall-clock entry gaps
def launches(r):
    o = np.argsort(r[:, 7]); r = r[o]
    cuts = [0] + [i for i in range(1, len(r)) if r[i, 7] - r[i - 1, 7] > 300] + [len(r)]       # > 3 us between entries = next launch
    return [r[i0:i1] for i0, i1 in zip(cuts[:-1], cuts[1:])]
for nm, r in (("k_dec_scores", s), ("k_dec_pv_stream", a)):
    ls = [x for x in launches(r) if len(x) > 8]
    span = np.median([(x[:, 6].max() - x[:, 7].min()) / 100.0 for x in ls])
    ent = np.median([(np.median(x[:, 7]) - x[:, 7].min()) / 100.0 for x in ls])
    entl = np.median([(x[:, 7].max() - x[:, 7].min()) / 100.0 for x in ls])
    life = np.median([np.median(x[:, 6] - x[:, 7]) / 100.0 for x in ls])
    print(f"{nm}: {len(ls)} launches, {int(np.median([len(x) for x in ls]))} workgroups each; first entry -> last exit {span:.2f} us; median entry +{ent:.2f}, last entry +{entl:.2f}; median workgroup lifetime {life:.2f} us")
if len(s):
    print(f"k_dec_scores workgroup (own clock): barrier passed +{us(np.median(s[:, 1] - s[:, 0])):.2f}, dots done +{us(np.median(s[:, 2] - s[:, 0])):.2f}")
m.close()

#!/usr/bin/env python3
"""In-kernel timeline of the persistent feed-forward launch (k_ffn_engine, ffn_engine.hip) inside the real decode loop.
Every workgroup records s_memtime at its phase boundaries (libllamahip_probe3.so, `make probe`): two records per workgroup from
the first consumer wave (kinds 0xE0 / 0xE1) and one from the loader wave (0xE2).  Stamps of one workgroup are differences on its own
clock; launches are lined up on the 100 MHz wall clock.
  consumers : entry | wo done (rows published) | edge 1 gathered | QA quantized | w1|w3 units done | activations published |
              edge 2 gathered (QA of w2 ready) | w2 done | exit
  loader    : entry | wo issued | w1|w3 issued | everything landed | cycles blocked on a full ring
usage: LLAMAHIP_LIB=libllamahip_probe3.so tools/engine_timeline.py [n_past] [steps]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("LLAMAHIP_LIB", "libllamahip_probe3.so")
os.environ.setdefault("LLAMAHIP_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L  # noqa: E402

import bench  # noqa: E402
model = os.environ.get("ENGINE_TIMELINE_MODEL", "7B")
path = bench.model_path(model, bench.MODELS[model], 20230312)
m = L.Model(path, n_ctx=512)
n_past = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prompt = (np.arange(n_past, dtype=np.int32) * 7919 + 13) % 32000
prompt[0] = 1
for c0 in range(0, n_past, 32):
    lg = m.eval(prompt[c0:c0 + 32], c0)
lib = L.lib()
lib.llamahip_debug_decode_phases.restype = C.c_int64
lib.llamahip_debug_decode_phases.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t]
cap = 120000 * steps
rec = np.zeros((cap, 8), np.uint64)
err = C.create_string_buffer(512)
n = lib.llamahip_debug_decode_phases(m._h, n_past, int(np.argmax(lg)), steps, rec.ctypes.data_as(C.c_void_p), cap, err, 512)
assert n > 0, err.value
rec = rec[:n].astype(np.int64)
kind = (rec[:, 5] >> 48) & 0xff
e0, e1, e2 = rec[kind == 0xE0], rec[kind == 0xE1], rec[kind == 0xE2]
if len(e0) == 0:
    print("no engine records: the engine did not run (LLAMAHIP_NO_ENGINE, or the probe library was not built with LH_PHASE_PROBE=3)")
    sys.exit(1)
# ticks per microsecond from the consumers' records (cycle span / wall span)
tpu = float((e1[:, 4] - e1[:, 0]).sum()) / (float((e1[:, 6] - e1[:, 7]).sum()) / 100.0)
# launches: consumer records clustered by wall-clock entry (launches of successive layers are > a few us apart in entry time)
order = np.argsort(e0[:, 7], kind="stable")
e0 = e0[order]
e1 = e1[np.argsort(e1[:, 7], kind="stable")]
e2 = e2[np.argsort(e2[:, 7], kind="stable")]
G = int(np.bincount((e0[:, 5] & 0xffffffff).astype(np.int64)).size)
nl = len(e0) // G
rows = []
for i in range(nl):
    a, b, ld = e0[i * G:(i + 1) * G], e1[i * G:(i + 1) * G], e2[i * G:(i + 1) * G]
    if len(a) < G or len(b) < G:
        continue
    t0 = a[:, 7].min() / 100.0
    own = lambda r, j: (r[:, j] - r[:, 0]) / tpu              # microseconds on the workgroup's own clock since its entry
    ent = a[:, 7] / 100.0 - t0
    row = dict(span=b[:, 6].max() / 100.0 - t0, entry_med=np.median(ent), entry_max=ent.max())
    for name, r, j in (("wo_done", a, 1), ("e1_gathered", a, 2), ("qa_done", a, 3), ("w13_done", a, 4), ("act_pub", b, 1), ("e2_gathered", b, 2), ("w2_done", b, 3), ("exit", b, 4)):
        v = ent + own(r, j)
        row[name + "_med"] = np.median(v); row[name + "_max"] = v.max(); row[name + "_min"] = v.min()
    if len(ld) == G:
        lent = ld[:, 7] / 100.0 - t0
        for name, j in (("ld_wo_issued", 1), ("ld_w13_issued", 2), ("ld_landed", 3)):
            v = lent + own(ld, j)
            row[name + "_med"] = np.median(v); row[name + "_max"] = v.max()
        row["ld_blocked_med"] = np.median(ld[:, 4] / tpu); row["ld_blocked_max"] = (ld[:, 4] / tpu).max()
    rows.append(row)
med = lambda f: float(np.median([r[f] for r in rows if f in r])) if any(f in r for r in rows) else float("nan")
print(f"{len(rows)} k_ffn_engine launches ({model}, n_past={n_past}, {G} workgroups); s_memtime = {tpu:.1f} ticks/us; medians over launches of the per-launch "
      f"median [min .. max] over workgroups, microseconds after the launch's first workgroup entered")
print(f"launch span {med('span'):.2f}   (workgroup entry: median {med('entry_med'):.2f}, last {med('entry_max'):.2f})")
for name, label in (("wo_done", "wo rows published"), ("e1_gathered", "edge 1 gathered (row h)"), ("qa_done", "norm -> Q4_0 done"), ("w13_done", "w1|w3 units done"),
                    ("act_pub", "activations quantized + published"), ("e2_gathered", "edge 2 gathered (QA of w2)"), ("w2_done", "w2 rows done"), ("exit", "exit")):
    print(f"  {label:36s} {med(name + '_med'):6.2f}   [{med(name + '_min'):6.2f} .. {med(name + '_max'):6.2f}]")
if any("ld_landed_med" in r for r in rows):
    print(f"  loader: wo issued {med('ld_wo_issued_med'):.2f} (last {med('ld_wo_issued_max'):.2f}) | w1|w3 issued {med('ld_w13_issued_med'):.2f} (last {med('ld_w13_issued_max'):.2f}) | "
          f"all landed {med('ld_landed_med'):.2f} (last {med('ld_landed_max'):.2f}) | blocked on a full ring {med('ld_blocked_med'):.2f} (max {med('ld_blocked_max'):.2f})")
e1c = med('e1_gathered_med') - med('wo_done_max')
e2c = med('e2_gathered_med') - med('act_pub_max')
print(f"edge cost (median gathered - LAST publisher): edge 1 {e1c:.2f} us, edge 2 {e2c:.2f} us;  edge 1 wait seen by the median workgroup {med('e1_gathered_med') - med('wo_done_med'):.2f}, edge 2 {med('e2_gathered_med') - med('act_pub_med'):.2f}")
m.close()

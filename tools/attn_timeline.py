#!/usr/bin/env python3
"""Timeline of the single-launch decode attention (k_dec_attn_x) in the real decode loop: every workgroup records
s_memtime at its phase boundaries (libllamahip_probe3.so, `make probe`).  All workgroups of a head sit on one XCD, so their
s_memtime stamps are comparable; launches are lined up on the 100 MHz wall clock.
  score workgroup  : entry | q roped (st[0], q, sin/cos loaded) | dots done, stores issued | stores acknowledged | arrived
  soft_max.V       : entry (V prefetch issued) | hand-off seen | soft_max done | V.P chains done | exit
usage: LLAMAHIP_LIB=libllamahip_probe3.so tools/attn_timeline.py [n_past] [steps]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("LLAMAHIP_LIB", "libllamahip_probe3.so")
os.environ.setdefault("LLAMAHIP_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_swift_amd as L  # noqa: E402

import bench  # noqa: E402
path = bench.model_path("7B", bench.MODELS["7B"], 20230312)          # (writes the synthetic model file if it is not there yet)
m = L.Model(path, n_ctx=512)
n_past = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prompt = (np.arange(n_past, dtype=np.int32) * 7919 + 13) % 32000
prompt[0] = 1
for c0 in range(0, n_past, 32):
    lg = m.eval(prompt[c0:c0 + 32], c0)
lib = L.lib()
lib.llamahip_debug_decode_phases.restype = C.c_int64
lib.llamahip_debug_decode_phases.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t]
cap = 80000 * steps
rec = np.zeros((cap, 8), np.uint64)
err = C.create_string_buffer(512)
n = lib.llamahip_debug_decode_phases(m._h, n_past, int(np.argmax(lg)), steps, rec.ctypes.data_as(C.c_void_p), cap, err, 512)
assert n > 0, err.value
rec = rec[:n].astype(np.int64)
kind = rec[:, 5] >> 48
# s_memtime ticks per microsecond.  A record's stamps 0 .. 4 lie INSIDE its wall-clock interval (the exit wall clock is read after the record's
# slot was claimed), so (stamp 4 - stamp 0) / (exit - entry) is a lower bound that the tightest records reach: take the 99.9th percentile
# (the all-records ratio of rounds 3-5 read 1 200 - 1 450 where tools/set_timeline.py, whose stamps bracket its wall clocks, measures ~2 300)
_w = (rec[:, 6] - rec[:, 7]) / 100.0
_ok = _w > 4.0
tpu = float(np.percentile((rec[_ok, 4] - rec[_ok, 0]) / _w[_ok], 99.9)) if _ok.any() else 2300.0
order = np.argsort(rec[:, 7], kind="stable")
rec = rec[order]; kind = kind[order]
# launches are serialised: a launch = a maximal run of records (by entry time) whose kinds all belong to the attention
# launch {mat-vec NORMP/STORE or NORM/STORE with 16 chunks, 0xA0 score, 0xA1 soft_max.V} or all do not
nch = (rec[:, 5] >> 32) & 0xffff
inatt = (kind == 0xA0) | (kind == 0xA1) | (kind == 0x44) | (kind == 0x24)
cuts = [0] + [i for i in range(1, len(rec)) if inatt[i] != inatt[i - 1]] + [len(rec)]
def absd(r, j):            # absolute time of stamp j on the wall clock, microseconds
    return r[:, 7] / 100.0 + (r[:, j] - r[:, 0]) / tpu
rows = []
for i0, i1 in zip(cuts[:-1], cuts[1:]):
    if not inatt[i0]:
        continue
    r = rec[i0:i1]; k = kind[i0:i1]
    rs = r[k == 0xA0]; rp = r[k == 0xA1]; rg = r[(k == 0x44) | (k == 0x24)]
    if len(rs) == 0 or len(rp) == 0:
        continue
    t0 = r[:, 7].min() / 100.0
    row = dict(span=r[:, 6].max() / 100.0 - t0, n_mv=len(rg), n_s=len(rs), n_p=len(rp))
    if len(rg):
        row.update(mv_entry=np.median(rg[:, 7]) / 100.0 - t0, mv_exit_med=np.median(rg[:, 6]) / 100.0 - t0, mv_exit_max=rg[:, 6].max() / 100.0 - t0,
                   mv_issue=np.median(rg[:, 1] - rg[:, 0]) / tpu, mv_prol=np.median(rg[:, 2] - rg[:, 0]) / tpu, mv_cons=np.median(rg[:, 3] - rg[:, 0]) / tpu, mv_wg=np.median(rg[:, 4] - rg[:, 0]) / tpu)
    row.update(s_entry=np.median(rs[:, 7]) / 100.0 - t0, s_roped=np.median(absd(rs, 1)) - t0, s_dots=np.median(absd(rs, 2)) - t0, s_ack=np.median(absd(rs, 3)) - t0,
               s_arr_med=np.median(absd(rs, 4)) - t0, s_arr_max=absd(rs, 4).max() - t0,
               p_entry=np.median(rp[:, 7]) / 100.0 - t0, p_seen=np.median(absd(rp, 1)) - t0, p_soft=np.median(absd(rp, 2)) - t0, p_chain=np.median(absd(rp, 3)) - t0,
               p_exit_med=np.median(rp[:, 6]) / 100.0 - t0, p_exit_max=rp[:, 6].max() / 100.0 - t0)
    rows.append(row)
med = lambda f: float(np.median([r[f] for r in rows if f in r])) if any(f in r for r in rows) else float("nan")
print(f"{len(rows)} attention launches at n_past={n_past}; s_memtime = {tpu:.1f} ticks/us; medians over launches, microseconds after the launch's first workgroup entered")
print(f"workgroups per launch: mat-vec {int(med('n_mv')) if rows and 'mv_entry' in rows[0] else 0}, score {int(med('n_s'))}, soft_max.V {int(med('n_p'))};  launch span {med('span'):.2f}")
if rows and "mv_entry" in rows[0]:
    print(f"mat-vec workgroup:    entry {med('mv_entry'):.2f} | (own clock: loads issued +{med('mv_issue'):.2f}, prologue done +{med('mv_prol'):.2f}, weights consumed +{med('mv_cons'):.2f}, stored +{med('mv_wg'):.2f}) | exit median {med('mv_exit_med'):.2f}, last {med('mv_exit_max'):.2f}")
print(f"score workgroup:      entry {med('s_entry'):.2f} | q roped {med('s_roped'):.2f} | dots done {med('s_dots'):.2f} | stores acknowledged {med('s_ack'):.2f} | arrived median {med('s_arr_med'):.2f}, last {med('s_arr_max'):.2f}")
print(f"soft_max.V workgroup: entry {med('p_entry'):.2f} | hand-off seen {med('p_seen'):.2f} | soft_max done {med('p_soft'):.2f} | V.P chains done {med('p_chain'):.2f} | exit median {med('p_exit_med'):.2f}, last {med('p_exit_max'):.2f}")
m.close()

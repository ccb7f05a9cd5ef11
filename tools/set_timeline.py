#!/usr/bin/env python3
"""In-kernel timeline of the few-row mat-mul (k_gemv_set) inside real set steps / 9-token evals: wave 0 of EVERY workgroup stamps
s_memtime at entry | ring issued | operands staged | {start, behind the barrier, items done} of its first four steps | loop done | exit
(libllamahip_setprobe.so = gemv_set.hip with -DLH_SET_PROBE=1, tools/build_set_variants.sh setprobe:"-DLH_SET_PROBE=1").  Records are
split into launches by the wall clock and summarised per kernel shape: medians over workgroups and launches, microseconds.
usage: tools/set_timeline.py [--seqs 4] [--steps 2] [--evals 9]"""
import argparse
import collections
import ctypes as C
import os
import sys

os.environ.setdefault("LLAMAHIP_LIB", "libllamahip_setprobe.so")
os.environ.setdefault("LLAMAHIP_SET_PROBE", "400000")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import llama_swift_amd as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="7B")
ap.add_argument("--seqs", type=int, default=4)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--evals", type=int, default=0)
ap.add_argument("--n_ctx", type=int, default=512)
args = ap.parse_args()
cfg = bench.MODELS[args.model]
path = bench.model_path(args.model, cfg, 20230312)
lib = L.lib()
lib.llamahip_debug_set_probe.restype = C.c_int64
lib.llamahip_debug_set_probe.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
rng = np.random.default_rng(5)
S = args.seqs
m = L.Model(path, n_ctx=args.n_ctx, n_seq=max(S, 1))
if args.evals:
    toks = rng.integers(3, cfg["n_vocab"], args.n_ctx).astype(np.int32)
    toks[0] = 1
    m.eval(toks[:args.evals], 0)
    lib.llamahip_debug_set_probe(None, 0, 1)
    for i in range(args.steps):
        m.eval(toks[(i + 1) * args.evals:(i + 2) * args.evals], (i + 1) * args.evals)
else:
    prompts = [np.concatenate([[1], rng.integers(3, cfg["n_vocab"], 7 + s)]).astype(np.int32) for s in range(S)]
    firsts = []
    for s in range(S):
        m.set_seq(s)
        firsts.append(int(np.argmax(m.eval(prompts[s], 0))))
    m.set_seq(0)
    bufs = [torch.tensor([firsts[s]], dtype=torch.int32, device="cuda") for s in range(S)]
    st = torch.cuda.current_stream().cuda_stream
    for s in range(S):
        m.stage_bind(s, len(prompts[s]), token_in=bufs[s].data_ptr(), token_out=bufs[s].data_ptr())
    for _ in range(3):
        m.stage_step_set(list(range(S)), 8, st)
    torch.cuda.synchronize()
    lib.llamahip_debug_set_probe(None, 0, 1)
    for _ in range(args.steps):
        m.stage_step_set(list(range(S)), 8, st)
    torch.cuda.synchronize()
cap = int(os.environ["LLAMAHIP_SET_PROBE"])
rec = np.zeros((cap, 32), np.uint64)
n = lib.llamahip_debug_set_probe(rec.ctypes.data_as(C.c_void_p), cap, 0)
assert n > 0, "no probe records: is LLAMAHIP_LIB a -DLH_SET_PROBE=1 build?"
rec = rec[:n].astype(np.int64)
rec = rec[np.argsort(rec[:, 0], kind="stable")]
ident = rec[:, 27] >> 24                       # (NC, CW, EPI, nchunks)
w0, w1 = rec[:, 0] / 100.0, rec[:, 26] / 100.0  # wall clock: 100 MHz
tpu = float((rec[:, 25] - rec[:, 1]).sum()) / max(1e-9, float((w1 - w0).sum()))      # s_memtime ticks per microsecond
launches, i = [], 0
while i < n:
    j, end = i + 1, w1[i]
    while j < n and ident[j] == ident[i] and w0[j] < end + 0.3:
        end = max(end, w1[j]); j += 1
    launches.append((i, j)); i = j
agg = collections.defaultdict(list)
for a, b in launches:
    r = rec[a:b]
    t0 = w0[a:b].min()
    d = lambda x, y: float(np.median(r[:, y] - r[:, x])) / tpu
    row = dict(nwg=b - a, ramp=w0[a:b].max() - t0, issue=d(1, 2), stage=d(2, 3), loop=d(3, 24), epi=d(24, 25), wg=d(1, 25),
               p50=float(np.median(w1[a:b])) - t0, span=w1[a:b].max() - t0)
    for s in range(4):
        ok = (r[:, 4 + 3 * s] > 0) & (r[:, 6 + 3 * s] > 0)
        if ok.any():
            rr = r[ok]
            nxt = rr[:, 4 + 3 * (s + 1)] if s < 3 else rr[:, 24]
            okn = nxt > 0
            row[f"s{s}_wait"] = float(np.median(rr[:, 5 + 3 * s] - rr[:, 4 + 3 * s])) / tpu
            row[f"s{s}_bar"] = float(np.median(rr[:, 6 + 3 * s] - rr[:, 5 + 3 * s])) / tpu
            if okn.any() and s < 3:
                row[f"s{s}_items"] = float(np.median(nxt[okn] - rr[okn, 6 + 3 * s])) / tpu
    agg[int(ident[a])].append(row)
epi = {0: "STORE", 1: "RESID", 3: "ROPE_KV", 7: "SILU_QAH"}
print(f"{n} workgroup records, {len(launches)} launches; s_memtime = {tpu:.1f} ticks/us; medians, microseconds")
print(f"{'kernel':30s} {'launches':>8s} {'WGs':>5s} {'ramp':>5s} | {'issue':>5s} {'stage':>6s} {'loop':>6s} {'epi':>5s} {'WG':>6s} | {'p50end':>6s} {'span':>6s} | per step: wait for own chunk / barrier / items")
for k, rows in sorted(agg.items()):
    nc, cw, ep, nch = (k >> 32) & 0xff, (k >> 24) & 0xff, (k >> 16) & 0xff, k & 0xffff
    med = lambda f: float(np.median([r[f] for r in rows if f in r])) if any(f in r for r in rows) else float("nan")
    steps = " ".join(f"{med(f's{s}_wait'):.2f}/{med(f's{s}_bar'):.2f}/{med(f's{s}_items'):.2f}" for s in range(3))
    print(f"<{nc},{cw},{epi.get(ep, ep)}> K={nch * 256:<6d}".ljust(30) + f" {len(rows):8d} {int(med('nwg')):5d} {med('ramp'):5.2f} | {med('issue'):5.2f} {med('stage'):6.2f} {med('loop'):6.2f} {med('epi'):5.2f} {med('wg'):6.2f} | "
          f"{med('p50'):6.2f} {med('span'):6.2f} | {steps}")
m.close()

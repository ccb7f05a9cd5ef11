"""One-shot GPU diagnostic: runs every parity check without stopping at the first failure and
prints where (which op / which intermediate) the HIP path first departs from the oracle.
Usage on the GPU box:  python tests/gpu_diag.py  [> gpurun_out/diag.log]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import llama_swift_amd as L  # noqa: E402
import reflib  # noqa: E402
import synth  # noqa: E402
from conftest import synth_tool  # noqa: E402

O = reflib.OracleLib()
fails = 0


def report(name, got, want):
    global fails
    got, want = np.asarray(got), np.asarray(want)
    if got.shape != want.shape:
        print(f"  FAIL {name}: shape {got.shape} vs {want.shape}"); fails += 1; return False
    if np.array_equal(got, want):
        print(f"  ok   {name} (bit-identical, n={got.size})"); return True
    g, w = got.astype(np.float64).ravel(), want.astype(np.float64).ravel()
    bad = np.flatnonzero(got.ravel() != want.ravel())
    print(f"  FAIL {name}: {bad.size}/{got.size} differ, max|d|={np.abs(g - w).max():.3e}, first idx {bad[:5]}, got {got.ravel()[bad[:3]]}, want {want.ravel()[bad[:3]]}")
    fails += 1
    return False


def main():
    print("version", L.version())
    rng = np.random.default_rng(0)
    # 1. activation quantizer
    print("[quantize_row]")
    x = rng.standard_normal(4096).astype(np.float32)
    x[32:64] = 0.0
    x[64:96] = np.linspace(-3.5, 3.5, 32, dtype=np.float32)       # ties at .5 after scaling
    report("quantize 4096", L.op_quantize_row_q4_0(x), O.quantize_row(x))
    # 2. mul_mat shapes
    print("[mul_mat_q4_0]")
    for (M, K, N) in [(8, 64, 1), (40, 256, 1), (64, 704, 2), (256, 4096, 1), (256, 4096, 9), (64, 11008, 1), (64, 11008, 5), (24, 5120, 3), (16, 8192, 1), (100, 4096, 17)]:
        w = synth.quantize_q4_0_offline((0.02 * rng.standard_normal((M, K))).astype(np.float32))
        xx = rng.standard_normal((N, K)).astype(np.float32)
        t0 = time.time()
        got = L.op_mul_mat_q4_0(w, xx)
        report(f"mul_mat M={M} K={K} N={N} ({time.time() - t0:.2f}s)", got, O.mul_mat_q4_0(w, xx, 4))
    # 3. tiny model, all intermediates
    print("[tiny model]")
    with tempfile.TemporaryDirectory() as td:
        hp = synth.HParams(n_vocab=96, n_embd=256, n_mult=64, n_head=2, n_layer=2)
        path = os.path.join(td, "tiny.bin")
        synth.write_model(path, hp, synth.random_tensors(hp, seed=5))
        for nth in (8, 3):
            om = O.load(path, 64)
            gm = L.Model(path, n_ctx=64)
            toks = synth.synth_prompt(9, hp.n_vocab, seed=2)
            for dl in (0, 1):
                a = gm.eval_debug(toks, 0, nth, dump_layer=dl) if dl == 0 else gm.eval_debug(np.array([7], np.int32), 9, nth, dump_layer=1)
                b = om.eval(toks, 0, nth, all_logits=True, dump_layer=dl) if dl == 0 else om.eval(np.array([7], np.int32), 9, nth, all_logits=True, dump_layer=1)
                for k in b:
                    report(f"nth={nth} dump_layer={dl} {k}", a[k], b[k])
            n_past = 10
            tok = 11
            for step in range(6):
                g = gm.eval(np.array([tok], np.int32), n_past, nth)
                w = om.eval(np.array([tok], np.int32), n_past, nth)["logits"]
                report(f"nth={nth} fused decode step {step}", g, w)
                tok = int(np.argmax(w)); n_past += 1
            for il in range(2):
                gk, gv = gm.kv(il, n_past); ok, ov = om.kv(il, n_past)
                report(f"nth={nth} K cache layer {il}", gk, ok); report(f"nth={nth} V cache layer {il}", gv, ov)
            first = tok
            for flags, label in ((1, "eager"), (3, "eager+unfused")):
                g2 = L.Model(path, n_ctx=64, flags=flags)
                g2.eval(toks, 0, nth); g2.eval(np.array([7], np.int32), 9, nth)
                tk = 11
                for step in range(6):
                    gl = g2.eval(np.array([tk], np.int32), 10 + step, nth)
                    wl = om2[step] if False else None
                    tk = int(np.argmax(gl))
                report(f"nth={nth} {label} greedy tokens vs graph model", g2.decode_greedy(first, n_past, 8, nth), gm.decode_greedy(first, n_past, 8, nth))
                g2.close()
            gt = gm.decode_greedy(first, n_past, 8, nth)
            ot = []
            for i in range(8):
                lo = om.eval(np.array([tok], np.int32), n_past + i, nth)["logits"]; tok = int(np.argmax(lo)); ot.append(tok)
            report(f"nth={nth} greedy tokens", gt, np.array(ot, np.int32))
            gm.close()
        # 4. medium model with the real 7B matrix shapes (2 layers)
        print("[7B-shaped 2-layer model]")
        mp = synth_tool(os.path.join(td, "m7.bin"), n_vocab=32000, n_embd=4096, n_mult=256, n_head=32, n_layer=2, seed=11)
        om = O.load(mp, 128); t0 = time.time(); gm = L.Model(mp, n_ctx=128); print(f"  load {time.time() - t0:.1f}s", gm.stats())
        toks = synth.synth_prompt(9, 32000, seed=4)
        a = gm.eval_debug(toks, 0, 8, dump_layer=1); b = om.eval(toks, 0, 8, all_logits=True, dump_layer=1)
        for k in b:
            report(f"7Bshape prompt {k}", a[k], b[k])
        tok = int(np.argmax(b["logits"])); n_past = 9
        for step in range(4):
            g = gm.eval(np.array([tok], np.int32), n_past, 8); w = om.eval(np.array([tok], np.int32), n_past, 8)["logits"]
            report(f"7Bshape fused decode step {step}", g, w); tok = int(np.argmax(w)); n_past += 1
        for which in range(5):
            r = gm.bench_gemv(which, 0, 5, 50)
            print(f"  bench {r['name']:9s} M={r['M']:6d} K={r['K']:6d} {r['us_per_launch']:8.2f} us  {r['GBps']:8.1f} GB/s  ({r['GBps'] / 8000 * 100:.1f}% of 8 TB/s)")
        t0 = time.time(); out = gm.decode_greedy(tok, n_past, 64, 8); dt = time.time() - t0
        print(f"  greedy decode 64 steps (2 layers): {dt * 1e3 / 64:.3f} ms/token")
        gm.close()
    print("FAILS:", fails)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())

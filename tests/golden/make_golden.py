#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REFERENCE ITSELF: the reference's own
Sources/cpp/ggml.c + utils.cpp compiled in place (oracle/_ref/libggml_ref.so, recipe in
oracle/Makefile) driven through oracle/ref_driver.cpp.  Run in the build container only (needs
/root/reference); the resulting .npz files are data (inputs + expected outputs) and are committed.

The reference has no tests or fixtures of its own, so these vectors are what pins
the standalone restatement oracle/oracle.c -- and, on the GPU box where /root/reference does not
exist, what the HIP path is compared with directly.

Compiler / flags of the reference build that produced them: gcc 11.4, -O3 -DNDEBUG -std=c11 -mavx
-mavx2 -mfma -mf16c -msse3 (tools/Makefile:34-36,78-99), glibc 2.35 libm, libstdc++ (sampler).
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import reflib  # noqa: E402
import synth  # noqa: E402

R = reflib.RefLib()
rng = np.random.default_rng(20230312)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  " + ", ".join(f"{k}{list(v.shape)}" for k, v in arrays.items()))


# ---------------------------------------------------------------- 1. Q4_0 block known-answer tests
def crafted_blocks():
    b = []
    b.append(np.zeros(32))                                         # all-zero block: d = 0, id = 0
    b.append(np.full(32, 1e-30))                                   # tiny but non-zero
    b.append(np.linspace(-3.5, 3.5, 32))                           # x*id lands exactly on .5 ties
    b.append(np.linspace(-7, 7, 32))
    halves = [i + 0.5 for i in range(-7, 7)] + [i + 0.5 for i in range(-7, 7)] + [0.5, -0.5, 2.5]     # 31 exact .5 ties, |x| <= 6.5
    b.append(np.array([7.0] + halves))                             # amax 7 -> id exactly 1: RNE vs half-away differ
    b.append(np.array([-7.0] + halves))
    b.append(np.array([3.5, -3.5, 2.5, -2.5, 1.5, -1.5, 0.5, -0.5] * 4) * 2.0)
    b.append(np.concatenate([np.full(31, 1e-3), [1e3]]))           # one outlier dominates
    b.append(np.concatenate([[-65504.0], np.full(31, 65504.0)]))
    b.append(np.full(32, -2.75))
    for e in (-20, -5, 0, 5, 20):
        b.append(rng.standard_normal(32) * (2.0 ** e))
    x = np.zeros(32); x[5] = 1.0; b.append(x)                      # single non-zero
    x = np.zeros(32); x[31] = -1.0; b.append(x)
    for k in range(8):                                             # values just around rounding boundaries
        base = (np.arange(32) % 15 - 7).astype(np.float64)
        b.append(base + (k - 4) * 1e-7 * 7)
    while len(b) < 64:
        b.append(rng.standard_normal(32) * rng.uniform(0.001, 10))
    return np.array(b, dtype=np.float32)


blocks = np.concatenate([crafted_blocks(), (rng.standard_normal((1000, 32)) * rng.uniform(0.01, 3, (1000, 1))).astype(np.float32)])
rt = np.stack([R.quantize_row(r) for r in blocks])                 # runtime quantizer (ggml.c:456-523)
off = R.quantize_offline(blocks).reshape(len(blocks), 20)          # offline quantizer (utils.cpp:431-485)
deq = np.stack([R.dequantize_row(r) for r in off])
save("q4_blocks.npz", x=blocks, runtime_q=rt, offline_q=off, dequant_of_offline=deq)

# ---------------------------------------------------------------- 2. mat-mul
mm = {}
for tag, (M, K, N) in {"a": (8, 64, 1), "b": (16, 256, 2), "c": (8, 4096, 9), "d": (8, 11008, 1), "e": (24, 704, 5)}.items():
    w = synth.quantize_q4_0_offline((0.02 * rng.standard_normal((M, K))).astype(np.float32))
    x = rng.standard_normal((N, K)).astype(np.float32)
    mm[f"{tag}_w"] = w
    mm[f"{tag}_x"] = x
    mm[f"{tag}_y"] = R.mul_mat_q4_0(w, x, 4)
    assert np.array_equal(mm[f"{tag}_y"], R.mul_mat_q4_0(w, x, 1))
save("mul_mat.npz", **mm)

# ---------------------------------------------------------------- 3. row ops
ops = {}
xn = (rng.standard_normal((5, 4096)) * 3 + 0.3).astype(np.float32)
ops["norm_x"], ops["norm_y"] = xn, R.unary_rows("norm", xn)
xs = np.concatenate([rng.standard_normal(2000) * 4, [0.0, -0.0, 1e-8, -1e-8, 30.0, -30.0, 70000.0, -70000.0, 11.09, -17.3]]).astype(np.float32).reshape(1, -1)
ops["silu_x"], ops["silu_y"] = xs, R.unary_rows("silu", xs)
xm = (rng.standard_normal((6, 77)) * 5).astype(np.float32)
xm[1, 40:] = -np.inf                                             # causal mask
xm[2, 1:] = -np.inf                                              # single visible key
xm[3] = 0.0
xm[4] -= 100.0
ops["softmax_x"], ops["softmax_y"] = xm, R.unary_rows("soft_max", xm)
xr = rng.standard_normal((7, 3, 128)).astype(np.float32)
ops["rope_x"] = xr
ops["rope_mode0_past5"] = R.rope(xr, 5, 0)
ops["rope_mode1_past4"] = R.rope(xr, 4, 1)
silu_in = np.array([R.h2f(i) for i in range(65536)], np.float32).reshape(1, -1)
finite = np.isfinite(silu_in)
silu_tab_in = np.where(finite, silu_in, 0).astype(np.float32)
silu_all = R.unary_rows("silu", silu_tab_in)
ops["silu_table_sha256"] = np.frombuffer(hashlib.sha256(silu_all.tobytes()).digest(), np.uint8)
save("ops.npz", **ops)

# ---------------------------------------------------------------- 4. tiny model, whole forward pass
hp = synth.HParams(n_vocab=96, n_embd=256, n_mult=64, n_head=2, n_layer=2)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "tiny.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=424242))
    file_bytes = np.fromfile(path, np.uint8)
    prompt = synth.synth_prompt(9, hp.n_vocab, seed=9)
    model = {"model_file": file_bytes, "prompt": prompt, "n_ctx": np.array([64])}
    for nth in (1, 8):
        m = R.load(path, 64)
        warm = m.eval(np.array([0, 1, 2, 3], np.int32), 0, nth)["logits"]        # the reference's warm-up eval (.mm:822)
        r = m.eval(prompt, 0, nth, all_logits=True, dump_layer=1)
        model[f"nth{nth}_warmup_logits"] = warm
        for k, v in r.items():
            model[f"nth{nth}_prompt_{k}"] = v
        tok, n_past, toks, lgs = int(np.argmax(r["logits"])), 9, [], []
        for _ in range(16):
            lg = m.eval(np.array([tok], np.int32), n_past, nth)["logits"]
            tok = int(np.argmax(lg)); n_past += 1
            toks.append(tok); lgs.append(lg)
        model[f"nth{nth}_greedy_tokens"] = np.array(toks, np.int32)
        model[f"nth{nth}_decode_logits"] = np.stack(lgs)
        k, v = m.kv(1, n_past)
        model[f"nth{nth}_kcache_l1"], model[f"nth{nth}_vcache_l1"] = k, v
        m.close()
    save("tiny_model.npz", **model)

    # ------------------------------------------------------------ 5. tokenizer + sampler (host logic)
    m = R.load(path, 64)
    prompts = ["abc", "hello world", " a b c", "tok00050tok00051x", "zzzz tok00095", "", "a", "ab tok0009", "the quick brown fox"]
    ts = {}
    for i, p in enumerate(prompts):
        ts[f"tok_{i}_bos"] = m.tokenize(p, True)
        ts[f"tok_{i}_nobos"] = m.tokenize(p, False)
    import ctypes as C
    L = R.L
    lg_seq = (rng.standard_normal((40, hp.n_vocab)) * 2.5).astype(np.float32)
    s = C.c_void_p(L.refllama_sampler_new(-1, 64))
    for t in prompt:
        L.refllama_sampler_accept(s, int(t))
    ids = []
    for i in range(40):
        tid = L.refllama_sampler_sample(m.h, s, np.ascontiguousarray(lg_seq[i]), 1.3, 40, float(np.float32(0.95)), float(np.float32(0.8)))
        L.refllama_sampler_accept(s, tid)
        ids.append(tid)
    L.refllama_sampler_free(s)
    ts["sampler_logits"] = lg_seq
    ts["sampler_window_init"] = prompt
    ts["sampler_ids"] = np.array(ids, np.int32)
    ts["prompts_joined"] = np.frombuffer(b"\x00".join(p.encode() for p in prompts), np.uint8)
    save("text.npz", **ts)
    m.close()
print("done")

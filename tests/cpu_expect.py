"""Child process of tests/test_gpu_fullsize.py: one CPU-side expectation (prompt logits, greedy tokens, final logits) written to an
.npz, so that the expectations of the full-size tests -- minutes of host time each -- are computed side by side while the GPU tests run
(the reference build keeps one static scratch buffer, .mm:529-547: two of its evals cannot share a process).
usage: cpu_expect.py <kind> <model file> <n_ctx> <n_prompt> <n_gen> <nth> <seed> <out.npz> [n_vocab = 32000]
  kind  decode   the reference path as the bridge drives it: 4-token scratch-sizing eval, ONE eval of the prompt, n_gen greedy tokens
        flow     the same with the prompt in the bridge's nine-token llama_eval calls (.mm:880-888)
        single   the standalone restatement (oracle.c): ONE eval of the prompt whatever its length (the reference's llama_eval cannot
                 take 2048 rows in one call), n_gen greedy tokens
        trace    ONE eval of the prompt (no scratch-sizing eval), then n_gen greedy tokens one llama_eval each, keeping every step's token
                 and top-2 logit margin (BASELINE.json configs[0] / [1]: 8-token prompt, 504 tokens to the end of the 512 context)
  <model file> may be "spec:<name>" (tests/bg_expect.py model(): written here if it is not on the box yet)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import reflib  # noqa: E402
import synth  # noqa: E402

kind, path, n_ctx, n_prompt, n_gen, nth, seed, out = sys.argv[1], sys.argv[2], *map(int, sys.argv[3:8]), sys.argv[8]
if path.startswith("spec:"):
    import bg_expect
    path = bg_expect.model(path[5:])
prompt = synth.synth_prompt(n_prompt, int(sys.argv[9]) if len(sys.argv) > 9 else 32000, seed=seed)
if kind == "single":
    os.environ.setdefault("ORC_OMP_THREADS", str(min(os.cpu_count() or 8, 64)))
    cpu = reflib.OracleLib().load(path, n_ctx)
else:
    lib = reflib.RefLib() if reflib.have_ref() else reflib.OracleLib()
    cpu = lib.load(path, n_ctx)          # (0 parts forced: the loader derives the part count from n_embd, .mm:33-38)
    if kind != "trace":
        cpu.eval(np.array([0, 1, 2, 3], np.int32), 0, nth)          # sizes the per-token scratch (.mm:820-822)
if kind == "flow":
    for c0 in range(0, n_prompt, 9):
        lg = cpu.eval(prompt[c0:c0 + 9], c0, nth)["logits"]
else:
    lg = cpu.eval(prompt, 0, nth)["logits"]
t, want, lo, margins = int(np.argmax(lg)), [], lg, []
for i in range(n_gen):
    lo = cpu.eval(np.array([t], np.int32), n_prompt + i, nth)["logits"]
    t = int(np.argmax(lo)); want.append(t)
    top2 = np.partition(lo, -2)[-2:]
    margins.append(float(top2[1] - top2[0]))
cpu.close()
np.savez(out + ".tmp.npz", prompt=prompt, lg=lg, first=int(np.argmax(lg)), want=np.array(want, np.int32), lo=lo, margins=np.array(margins))
os.replace(out + ".tmp.npz", out)

"""Full-depth parity of the other BASELINE.json configurations against the CPU path (the reference's own ggml.c build,
oracle/_ref, when it travelled with the snapshot; else the standalone restatement):

  configs[2]  LLaMA-7B, all 32 layers, a 2048-token prompt in ONE eval at n_ctx 2560 (the eval bench.py's prefill leg times)
  configs[3]  LLaMA-13B, 40 layers, 2-part file (.mm:33-38, merged as .mm:312-495)
  configs[4]  LLaMA-65B, 80 layers, 8-part file -- the model the 8-GPU pipeline shards

Model files are synthetic (random Q4_0 weights of the exact shapes, written in the reference's file format by
csrc/tools/make_synth_model) and shared with bench.py through LLAMAHIP_MODEL_DIR.  The 65B file is 40 GB: that test
needs ~45 GB of /tmp and of host RAM and takes a few minutes; LLAMAHIP_SKIP_65B=1 skips it.
"""
import os

import numpy as np
import pytest

import synth
from conftest import synth_tool

pytestmark = pytest.mark.gpu


def same(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def describe(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return f"shape {a.shape} vs {b.shape}"
    bad = np.flatnonzero(a.ravel() != b.ravel())
    return f"{bad.size}/{a.size} differ, first at {bad[:4]}, got {a.ravel()[bad[:3]]} want {b.ravel()[bad[:3]]}"


def _model(preset: str) -> str:
    d = os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models")
    path = os.path.join(d, f"{preset}-seed20230312", "ggml-model-q4_0.bin")
    if not os.path.exists(path + ".done"):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        synth_tool(path, preset=preset, seed=20230312)
        open(path + ".done", "w").close()
    return path


def _cpu_lib():
    import reflib
    return reflib.RefLib() if reflib.have_ref() else reflib.OracleLib()


def _cpu_load(path, n_ctx, nth):
    """The CPU path as the bridge drives it: load, then the 4-token eval that sizes its per-token scratch (.mm:820-822) -- without it
    the reference's fixed 512 MiB eval buffer (.mm:529-547) overflows on 65B-sized or 2048-token evals."""
    cpu = _cpu_lib().load(path, n_ctx)          # (0 parts forced: the loader derives the part count from n_embd, .mm:33-38)
    cpu.eval(np.array([0, 1, 2, 3], np.int32), 0, nth)
    return cpu


def _decode_vs_cpu(L, path, n_ctx, n_prompt, n_gen, nth=8):
    """prompt eval (last-row logits bit for bit), then n_gen greedy tokens: the device-resident loop against one CPU eval per
    token, final logits bit for bit, and the same tokens once more through one host-driven llamahip_eval per token."""
    cpu = _cpu_load(path, n_ctx, nth)
    prompt = synth.synth_prompt(n_prompt, 32000, seed=3)
    lg = cpu.eval(prompt, 0, nth)["logits"]
    first, want, t = int(np.argmax(lg)), [], None
    t = first
    for i in range(n_gen):
        lo = cpu.eval(np.array([t], np.int32), n_prompt + i, nth)["logits"]
        t = int(np.argmax(lo)); want.append(t)
    cpu.close()
    with L.Model(path, n_ctx=n_ctx) as gm:
        a = gm.eval(prompt, 0, nth)
        assert same(a, lg), "prompt logits: " + describe(a, lg)
        got, last = gm.decode_greedy(first, n_prompt, n_gen, nth, want_logits=True)
        assert got.tolist() == want, (got.tolist(), want)
        assert same(last, lo), "final logits: " + describe(last, lo)
        t, got2 = first, []
        for i in range(n_gen):
            lg2 = gm.eval(np.array([t], np.int32), n_prompt + i, nth)
            t = int(np.argmax(lg2)); got2.append(t)
        assert got2 == want and same(lg2, lo)


def test_13b_full_depth_vs_cpu_path(L):
    """configs[3]: 40 layers, n_embd 5120, two-part file; 9-token prompt + 16 greedy tokens."""
    _decode_vs_cpu(L, _model("13B"), 128, 9, 16)


@pytest.mark.skipif(os.environ.get("LLAMAHIP_SKIP_65B") == "1", reason="LLAMAHIP_SKIP_65B=1")
def test_65b_full_depth_vs_cpu_path(L):
    """configs[4]'s model on one GPU: 80 layers, n_embd 8192, eight-part file; 9-token prompt + 4 greedy tokens."""
    _decode_vs_cpu(L, _model("65B"), 64, 9, 4)


def test_7b_full_depth_2048_token_prefill_vs_cpu_path(L):
    """configs[2]: the 32-layer 2048-token single eval that bench.py's prefill leg times (matrix-core GEMMs, lane-per-query
    attention), last-row logits bit for bit, then 3 decode tokens from that context.  The CPU side is the standalone restatement
    (oracle/oracle.c, pinned against the reference build by tests/test_oracle_vs_ref.py): the reference's own llama_eval cannot take
    2048 tokens in one call -- its scratch buffer is sized from a 4-token eval and the attention scores grow with N^2 (.mm:529-547,
    727-729), which is why the bridge feeds prompts 8 tokens at a time -- and a chunked evaluation is a different computation (the
    V*P key split depends on the keys of the eval, ggml.c:5619-5665).  The restatement's row / (head, query) loops run on all host
    cores (ORC_OMP_THREADS; n_threads = 8 stays the arithmetic's parameter)."""
    import reflib
    os.environ.setdefault("ORC_OMP_THREADS", str(min(os.cpu_count() or 8, 64)))
    path = _model("7B")
    prompt = synth.synth_prompt(2048, 32000, seed=5)
    cpu = reflib.OracleLib().load(path, 2560)
    lg = cpu.eval(prompt, 0, 8)["logits"]
    t, want = int(np.argmax(lg)), []
    first = t
    for i in range(3):
        lo = cpu.eval(np.array([t], np.int32), 2048 + i, 8)["logits"]
        t = int(np.argmax(lo)); want.append(t)
    cpu.close()
    with L.Model(path, n_ctx=2560) as gm:
        a = gm.eval(prompt, 0, 8)
        assert same(a, lg), "last-row logits of the 2048-token eval: " + describe(a, lg)
        got, last = gm.decode_greedy(first, 2048, 3, 8, want_logits=True)
        assert got.tolist() == want and same(last, lo), (got.tolist(), want, describe(last, lo))


def test_7b_full_depth_2048_token_prompt_in_the_reference_flow_vs_reference(L):
    """configs[2] the way the REFERENCE evaluates a 2048-token prompt -- the bridge's loop of nine-token llama_eval calls behind its
    4-token warm-up (.mm:820-822, 840-848, 880-888), 228 evals of the reference's own ggml.c on the host -- against ONE
    llamahip_eval_chunks pass over the 2048 rows on the device: final logits bit for bit, then 3 greedy tokens (the long-context
    decode schedule) against one reference eval each, final logits bit for bit."""
    path = _model("7B")
    prompt = synth.synth_prompt(2048, 32000, seed=6)
    cpu = _cpu_load(path, 2560, 8)
    for c0 in range(0, 2048, 9):
        lg = cpu.eval(prompt[c0:c0 + 9], c0, 8)["logits"]
    t, want = int(np.argmax(lg)), []
    first = t
    for i in range(3):
        lo = cpu.eval(np.array([t], np.int32), 2048 + i, 8)["logits"]
        t = int(np.argmax(lo)); want.append(t)
    cpu.close()
    with L.Model(path, n_ctx=2560) as gm:
        gm.eval(np.array([0, 1, 2, 3], np.int32), 0, 8)
        a = gm.eval_chunks(prompt, 0, 9, 8)
        assert same(a, lg), "logits after the 2048-token prompt: " + describe(a, lg)
        got, last = gm.decode_greedy(first, 2048, 3, 8, want_logits=True)
        assert got.tolist() == want and same(last, lo), (got.tolist(), want, describe(last, lo))

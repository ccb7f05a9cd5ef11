"""Full-depth parity of the other BASELINE.json configurations against the CPU path (the reference's own ggml.c build,
oracle/_ref, when it travelled with the snapshot; else the standalone restatement):

  configs[2]  LLaMA-7B, all 32 layers, a 2048-token prompt in ONE eval at n_ctx 2560 (the eval bench.py's prefill leg times)
  configs[3]  LLaMA-13B, 40 layers, 2-part file (.mm:33-38, merged as .mm:312-495)
  configs[4]  LLaMA-65B, 80 layers, 8-part file -- the model the 8-GPU pipeline shards

Model files are synthetic (random Q4_0 weights of the exact shapes, written in the reference's file format by
csrc/tools/make_synth_model) and shared with bench.py through LLAMAHIP_MODEL_DIR.  The CPU expectations are child processes
started when the session is collected (tests/bg_expect.py): they compute side by side while earlier GPU tests run.  The 65B file is 40 GB: that test
needs ~45 GB of /tmp and of host RAM and takes a few minutes; LLAMAHIP_SKIP_65B=1 skips it.
"""
import os

import numpy as np
import pytest

import synth

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


import bg_expect                                  # model files + background CPU expectations (started at collection, conftest.py)

_model = bg_expect.model
_WIDE = bg_expect.WIDE


class _Expect:
    get = staticmethod(bg_expect.get)


@pytest.fixture(scope="module")
def expect():
    return _Expect()


def _decode_vs_cpu(L, expect, name, preset, n_ctx, n_prompt, n_gen, nth=8):
    """prompt eval (last-row logits bit for bit), then n_gen greedy tokens: the device-resident loop against one CPU eval per
    token, final logits bit for bit."""
    x = expect.get(name)
    path = _model(preset)
    prompt, lg, first, want, lo = x["prompt"], x["lg"], int(x["first"]), x["want"].tolist(), x["lo"]
    assert len(prompt) == n_prompt and len(want) == n_gen
    with L.Model(path, n_ctx=n_ctx) as gm:
        a = gm.eval(prompt, 0, nth)
        assert same(a, lg), "prompt logits: " + describe(a, lg)
        got, last = gm.decode_greedy(first, n_prompt, n_gen, nth, want_logits=True)
        assert got.tolist() == want, (got.tolist(), want)
        assert same(last, lo), "final logits: " + describe(last, lo)


def test_13b_full_depth_vs_cpu_path(L, expect):
    """configs[3]: 40 layers, n_embd 5120, two-part file; 9-token prompt (the reference's first prompt batch) + 5 greedy tokens."""
    _decode_vs_cpu(L, expect, "13B", "13B", 64, 9, 5)


@pytest.mark.skipif(bool(os.environ.get("LLAMAHIP_SKIP_65B")), reason="LLAMAHIP_SKIP_65B set")
def test_65b_full_depth_vs_cpu_path(L, expect):
    """configs[4]'s model on one GPU: 80 layers, n_embd 8192, eight-part file; 9-token prompt + 4 greedy tokens."""
    _decode_vs_cpu(L, expect, "65B", "65B", 64, 9, 4)


def test_7b_full_depth_2048_token_prefill_vs_cpu_path(L, expect):
    """configs[2]: the 32-layer 2048-token single eval that bench.py's prefill leg times (matrix-core GEMMs, lane-per-query
    attention), last-row logits bit for bit, then 3 decode tokens from that context.  The CPU side is the standalone restatement
    (oracle/oracle.c, pinned against the reference build by tests/test_oracle_vs_ref.py): the reference's own llama_eval cannot take
    2048 tokens in one call -- its scratch buffer is sized from a 4-token eval and the attention scores grow with N^2 (.mm:529-547,
    727-729), which is why the bridge feeds prompts 8 tokens at a time -- and a chunked evaluation is a different computation (the
    V*P key split depends on the keys of the eval, ggml.c:5619-5665; next test).  The restatement's row / (head, query) loops run on
    all host cores (ORC_OMP_THREADS; n_threads = 8 stays the arithmetic's parameter)."""
    _decode_vs_cpu(L, expect, "single2048", "7B", 2560, 2048, 3)


def test_7b_full_depth_2048_token_prompt_in_the_reference_flow_vs_reference(L, expect):
    """configs[2] the way the REFERENCE evaluates a 2048-token prompt -- the bridge's loop of nine-token llama_eval calls behind its
    4-token warm-up (.mm:820-822, 840-848, 880-888), 228 evals of the reference's own ggml.c on the host -- against ONE
    llamahip_eval_chunks pass over the 2048 rows on the device: final logits bit for bit, then 3 greedy tokens (the long-context
    decode schedule) against one reference eval each, final logits bit for bit."""
    x = expect.get("flow2048")
    prompt, lg, first, want, lo = x["prompt"], x["lg"], int(x["first"]), x["want"].tolist(), x["lo"]
    with L.Model(_model("7B"), n_ctx=2560) as gm:
        gm.eval(np.array([0, 1, 2, 3], np.int32), 0, 8)
        a = gm.eval_chunks(prompt, 0, 9, 8)
        assert same(a, lg), "logits after the 2048-token prompt: " + describe(a, lg)
        got, last = gm.decode_greedy(first, 2048, 3, 8, want_logits=True)
        assert got.tolist() == want and same(last, lo), (got.tolist(), want, describe(last, lo))


def _flow_vs_cpu(L, expect, name, path, n_ctx, n_gen):
    """the prompt in the bridge's nine-token evals (CPU: the reference's own llama_eval calls; device: ONE llamahip_eval_chunks pass),
    then n_gen greedy tokens through the decode step's production schedule selection: tokens and final logits bit for bit."""
    x = expect.get(name)
    prompt, lg, first, want, lo = x["prompt"], x["lg"], int(x["first"]), x["want"].tolist(), x["lo"]
    assert len(want) == n_gen
    with L.Model(path, n_ctx=n_ctx) as gm:
        gm.eval(np.array([0, 1, 2, 3], np.int32), 0, 8)
        a = gm.eval_chunks(prompt, 0, 9, 8)
        assert same(a, lg), "logits after the prompt: " + describe(a, lg)
        got, last = gm.decode_greedy(first, len(prompt), n_gen, 8, want_logits=True)
        assert got.tolist() == want, (name, got.tolist(), want)
        assert same(last, lo), "final logits: " + describe(last, lo)


def test_13b_full_depth_128_token_prompt_and_32_tokens_vs_reference(L, expect):
    """configs[3] beyond position 14: all 40 layers, a 128-token prompt in the reference's flow (fifteen llama_eval calls of the
    reference build) against one llamahip_eval_chunks pass, then 32 greedy tokens (positions 128 .. 159)."""
    _flow_vs_cpu(L, expect, "13B_128", _model("13B"), 256, 32)


@pytest.mark.parametrize("name", sorted(_WIDE))
def test_wide_models_decode_across_their_default_attention_schedule_thresholds(L, expect, name):
    """13B-width and 65B-width rows decoded ACROSS the positions where attn_sched_at switches the decode step's attention schedule
    (no environment override: the thresholds the production build uses), against the reference build."""
    kw, n_ctx, n_prompt, n_gen = _WIDE[name]
    _flow_vs_cpu(L, expect, name, _model(name.split("_")[0]), n_ctx, n_gen)

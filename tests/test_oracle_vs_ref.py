"""CPU: the standalone restatement against the reference's own ggml.c compiled in place
(oracle/_ref, present in the build container and shipped prebuilt to the GPU box).  Randomised,
larger shapes than the committed fixtures.  Bit-exact."""
import numpy as np
import pytest

import synth


def test_quantizers(oracle, ref):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((64, 4096)) * rng.uniform(0.01, 5, (64, 1))).astype(np.float32)
    x[3, 64:96] = 0
    for r in x[:16]:
        assert np.array_equal(oracle.quantize_row(r), ref.quantize_row(r))
    assert np.array_equal(oracle.quantize_offline(x), ref.quantize_offline(x))
    assert np.array_equal(synth.quantize_q4_0_offline(x), ref.quantize_offline(x))      # the numpy writer too


@pytest.mark.parametrize("M,K,N", [(64, 4096, 1), (32, 11008, 3), (40, 5120, 9), (16, 8192, 2), (128, 64, 4)])
def test_mul_mat(oracle, ref, M, K, N):
    rng = np.random.default_rng(M * 7 + K + N)
    w = synth.quantize_q4_0_offline((0.02 * rng.standard_normal((M, K))).astype(np.float32))
    x = rng.standard_normal((N, K)).astype(np.float32)
    assert np.array_equal(oracle.mul_mat_q4_0(w, x, 4), ref.mul_mat_q4_0(w, x, 8))


def test_row_ops(oracle, ref):
    rng = np.random.default_rng(2)
    x = (rng.standard_normal((9, 5120)) * 2).astype(np.float32)
    assert np.array_equal(oracle.unary_rows("norm", x), ref.unary_rows("norm", x, 4))
    assert np.array_equal(oracle.unary_rows("silu", x), ref.unary_rows("silu", x, 4))
    s = (rng.standard_normal((40, 513)) * 4).astype(np.float32)
    s[::3, 200:] = -np.inf
    assert np.array_equal(oracle.unary_rows("soft_max", s), ref.unary_rows("soft_max", s, 4))
    r = rng.standard_normal((12, 4, 128)).astype(np.float32)
    assert np.array_equal(oracle.rope(r, 37, 0), ref.rope(r, 37, 0))
    assert np.array_equal(oracle.rope(r, 5, 1), ref.rope(r, 5, 1))


@pytest.mark.parametrize("nth,parts", [(8, 1), (5, 2)])
def test_model_eval_and_multipart_merge(oracle, ref, tmp_path, nth, parts):
    hp = synth.HParams(n_vocab=128, n_embd=256, n_mult=256, n_head=2, n_layer=3)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=77), n_parts=parts)
    mo, mr = oracle.load(path, 48, parts), ref.load(path, 48, parts)
    for name in ("tok_embeddings.weight", "layers.1.attention.wo.weight", "layers.2.feed_forward.w2.weight",
                 "layers.0.feed_forward.w1.weight", "output.weight", "norm.weight"):
        assert np.array_equal(mo.tensor_bytes(name), mr.tensor_bytes(name)), name
    toks = synth.synth_prompt(11, hp.n_vocab, seed=5)
    a, b = mo.eval(toks[:9], 0, nth, all_logits=True, dump_layer=2), mr.eval(toks[:9], 0, nth, all_logits=True, dump_layer=2)
    for k in b:
        assert np.array_equal(a[k], b[k]), k
    a, b = mo.eval(toks[9:], 9, nth, all_logits=True), mr.eval(toks[9:], 9, nth, all_logits=True)
    assert np.array_equal(a["logits_all"], b["logits_all"])
    tok, n_past = int(np.argmax(b["logits"])), 11
    for _ in range(10):
        la = mo.eval(np.array([tok], np.int32), n_past, nth)["logits"]
        lb = mr.eval(np.array([tok], np.int32), n_past, nth)["logits"]
        assert np.array_equal(la, lb)
        tok = int(np.argmax(lb)); n_past += 1

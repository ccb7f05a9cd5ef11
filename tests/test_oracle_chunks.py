"""CPU: the claim behind llamahip_eval_chunks, checked on the CPU side alone.  The reference's driver hands a prompt to llama_eval
n_batch + 1 = 9 tokens at a time (.mm:840-848, 880-888).  Every operator of the eval graph works row by row except the V*P key split over
n_threads, whose chunk length comes from the key count OF THAT llama_eval CALL (ggml.c:5459-5480).  So ONE pass over all prompt rows in
which row n splits n_past + min(N, (n // 9 + 1) * 9) keys must leave the same KV cache and logits, bit for bit, as the successive calls
-- here: the reference build's successive calls (oracle/_ref, the reference's own ggml.c) against the restatement's one pass with
orc_set_split_chunk(9) -- and an ordinary one-call eval of the same rows must NOT (else the test would say nothing)."""
import numpy as np
import pytest

import synth


@pytest.mark.parametrize("nth,chunk,n_prompt,n_head", [(8, 9, 40, 2), (3, 9, 100, 2), (5, 4, 30, 4), (1, 9, 25, 2)])
def test_one_pass_with_per_row_key_split_equals_successive_calls(oracle, ref, tmp_path, nth, chunk, n_prompt, n_head):
    hp = synth.HParams(n_vocab=96, n_embd=256, n_mult=64, n_head=n_head, n_layer=2)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=31))
    prompt = synth.synth_prompt(n_prompt, hp.n_vocab, seed=8)
    warm = np.array([0, 1, 2, 3], np.int32)
    n_ctx, n_past = n_prompt + 8, 4
    rm = ref.load(path, n_ctx)
    rm.eval(warm, 0, nth)
    for c0 in range(0, n_prompt, chunk):
        want = rm.eval(prompt[c0:c0 + chunk], n_past + c0, nth)["logits"]
    om = oracle.load(path, n_ctx)
    om.eval(warm, 0, nth)
    oracle.L.orc_set_split_chunk(chunk)
    try:
        got = om.eval(prompt, n_past, nth)["logits"]
    finally:
        oracle.L.orc_set_split_chunk(0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    T = n_past + n_prompt
    for il in range(hp.n_layer):
        gk, gv = om.kv(il, T)
        rk, rv = rm.kv(il, T)
        assert np.array_equal(gk.view(np.uint32), rk.view(np.uint32)) and np.array_equal(gv.view(np.uint32), rv.view(np.uint32)), f"kv cache layer {il}"
    if nth > 1:
        o1 = oracle.load(path, n_ctx)
        o1.eval(warm, 0, nth)
        one = o1.eval(prompt, n_past, nth)["logits"]
        assert not np.array_equal(one.view(np.uint32), want.view(np.uint32))

"""bench.py's own logic, without a GPU: the parser that turns a rocprofv3 kernel trace of the decode loop into the in-situ
per-launch table `roofline` is computed from must recognise every launch layout the library can run."""
import csv
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CFG = dict(bench.MODELS["7B"], n_layer=3)          # three layers keep the synthetic traces short

LAYOUTS = {
    "fused": ["lh::k_qkv_attn<4, 8, 1, false>(args)", "void lh::k_gemv<0, 1, 16, false, 4>(a)", "void lh::k_gemv<4, 2, 4, true, 1>(a)", "void lh::k_gemv<0, 1, 10, true, 12>(a)"],
    "attn_x": ["void lh::k_gemv<4, 0, 8, true, 1>(a)", "lh::k_dec_attn_x(a)", "void lh::k_gemv<0, 1, 16, false, 4>(a)", "void lh::k_gemv<4, 2, 4, true, 1>(a)",
               "void lh::k_gemv<0, 1, 10, true, 12>(a)"],
    "two": ["void lh::k_gemv<4, 0, 8, true, 1>(a)", "lh::k_dec_scores(a)", "void lh::k_dec_pv_blk<false>(a)", "void lh::k_gemv<0, 1, 16, false, 4>(a)",
            "void lh::k_gemv<4, 2, 4, true, 1>(a)", "void lh::k_gemv<0, 1, 10, true, 12>(a)"],
    "stream": ["void lh::k_gemv<4, 0, 8, true, 1>(a)", "lh::k_dec_scores(a)", "void lh::k_dec_pv_stream<4>(a)", "void lh::k_gemv<0, 1, 16, false, 4>(a)",
               "void lh::k_gemv<4, 2, 4, true, 1>(a)", "void lh::k_gemv<0, 1, 10, true, 12>(a)"],
}


def write_trace(path, layer, tokens):
    t = 1000
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Agent_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writerow(["KERNEL_DISPATCH", 1, "lh::k_repack_q4(a)", t, t + 500]); t += 1000          # load-time noise before the loop
        for _ in range(tokens):
            seq = ["lh::k_embed_part(a)"] + layer * CFG["n_layer"] + ["void lh::k_gemv<4, 0, 4, true, 1>(a)", "void lh::k_argmax<1>(a)"]
            for i, name in enumerate(seq):
                dur = 1000 * (1 + i % 7)
                w.writerow(["KERNEL_DISPATCH", 1, name, t, t + dur]); t += dur + 300


@pytest.mark.parametrize("kind", sorted(LAYOUTS))
def test_insitu_trace_parser_knows_every_decode_layout(tmp_path, kind):
    d = tmp_path / kind
    d.mkdir()
    write_trace(str(d / "x_kernel_trace.csv"), LAYOUTS[kind], tokens=5)
    prof = bench.parse_kernel_trace(str(d), CFG)
    assert prof and prof["tokens"] == 5
    roles = set(prof["us"])
    assert {"w1|w3", "w2", "wo", "output", "embed", "argmax", "token_span"} <= roles
    if kind == "fused":
        assert "wq|wk|wv+attention" in roles and "k_qkv_attn" in prof["kernel"]["wq|wk|wv+attention"]
    elif kind == "attn_x":
        assert {"wq|wk|wv", "attention"} <= roles
    else:
        assert {"wq|wk|wv", "attn_scores", "attn_softmax_pv"} <= roles
        assert ("k_dec_pv_stream" if kind == "stream" else "k_dec_pv_blk") in prof["kernel"]["attn_softmax_pv"]
    assert "k_gemv<4, 2, 4" in prof["kernel"]["w1|w3"]


def test_insitu_trace_parser_rejects_a_foreign_sequence(tmp_path):
    d = tmp_path / "bad"
    d.mkdir()
    write_trace(str(d / "x_kernel_trace.csv"), ["lh::k_something_else(a)"] * 4, tokens=3)
    assert bench.parse_kernel_trace(str(d), CFG) is None


def test_token_bytes_follow_the_survey_formula():
    cfg = bench.MODELS["7B"]
    d, F, V, Lr = cfg["n_embd"], bench.n_ff(cfg), cfg["n_vocab"], cfg["n_layer"]
    W = (Lr * (4 * d * d + 3 * d * F) + V * d) // 32 * 20
    assert abs(W - 4.129e9) / 4.129e9 < 0.01                      # 4.129 GB of Q4_0 weights per 7B token
    assert bench.token_bytes(cfg, 100) - bench.token_bytes(cfg, 99) == Lr * 2 * d * 4          # one more K and V row per layer

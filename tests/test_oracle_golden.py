"""CPU: the standalone restatement (oracle/oracle.c) against the committed golden vectors, which were
produced by the reference's own ggml.c compiled in place (tests/golden/make_golden.py).  Bit-exact."""
import hashlib
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_q4_block_kats(oracle):
    g = load("q4_blocks.npz")
    x = g["x"]
    for i in range(len(x)):
        assert np.array_equal(oracle.quantize_row(x[i]), g["runtime_q"][i]), f"runtime quantizer, block {i}"
    assert np.array_equal(oracle.quantize_offline(x).reshape(len(x), 20), g["offline_q"])
    for i in range(0, len(x), 7):
        assert np.array_equal(oracle.dequantize_row(g["offline_q"][i]), g["dequant_of_offline"][i])


def test_runtime_and_offline_quantizers_really_differ():
    # ties at .5 round to even at run time (AVX2 branch) and away from zero offline: the KATs must
    # contain blocks that tell the two apart, otherwise they pin nothing
    g = load("q4_blocks.npz")
    assert (g["runtime_q"] != g["offline_q"]).any()


@pytest.mark.parametrize("tag", list("abcde"))
def test_mul_mat(oracle, tag):
    g = load("mul_mat.npz")
    y = oracle.mul_mat_q4_0(g[f"{tag}_w"], g[f"{tag}_x"], 3)
    assert np.array_equal(y, g[f"{tag}_y"])


def test_vec_dot_scalar_equals_simd(oracle):
    g = load("mul_mat.npz")
    w, x = g["c_w"], g["c_x"]
    qa = oracle.quantize_row(x[0])
    for m in range(w.shape[0]):
        assert oracle.vec_dot_q4_0(w[m], qa, scalar=True) == oracle.vec_dot_q4_0(w[m], qa, scalar=False)
        assert oracle.vec_dot_q4_0(w[m], qa) == g["c_y"][0, m]


def test_row_ops(oracle):
    g = load("ops.npz")
    assert np.array_equal(oracle.unary_rows("norm", g["norm_x"]), g["norm_y"])
    # compare bit patterns: silu(-inf) is NaN in the reference (-inf / inf), and NaN != NaN
    assert np.array_equal(oracle.unary_rows("silu", g["silu_x"]).view(np.uint32), g["silu_y"].view(np.uint32))
    assert np.array_equal(oracle.unary_rows("soft_max", g["softmax_x"]), g["softmax_y"])
    assert np.array_equal(oracle.rope(g["rope_x"], 5, 0), g["rope_mode0_past5"])
    assert np.array_equal(oracle.rope(g["rope_x"], 4, 1), g["rope_mode1_past4"])


def test_silu_table(oracle):
    g = load("ops.npz")
    silu, _ = oracle.tables()
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    x = np.where(np.isfinite(h), h, 0).astype(np.float32).reshape(1, -1)
    y = oracle.unary_rows("silu", x)
    assert hashlib.sha256(y.tobytes()).digest() == g["silu_table_sha256"].tobytes()
    assert silu.dtype == np.uint16 and silu.size == 65536


@pytest.mark.parametrize("nth", [1, 8])
def test_tiny_model_forward(oracle, tmp_path, nth):
    g = load("tiny_model.npz")
    path = str(tmp_path / "tiny.bin")
    g["model_file"].tofile(path)
    m = oracle.load(path, int(g["n_ctx"][0]))
    assert np.array_equal(m.eval(np.array([0, 1, 2, 3], np.int32), 0, nth)["logits"], g[f"nth{nth}_warmup_logits"])
    r = m.eval(g["prompt"], 0, nth, all_logits=True, dump_layer=1)
    for k, v in r.items():
        assert np.array_equal(v, g[f"nth{nth}_prompt_{k}"]), k
    tok, n_past = int(np.argmax(r["logits"])), 9
    for i in range(16):
        lg = m.eval(np.array([tok], np.int32), n_past, nth)["logits"]
        assert np.array_equal(lg, g[f"nth{nth}_decode_logits"][i]), f"decode step {i}"
        tok = int(np.argmax(lg)); n_past += 1
        assert tok == g[f"nth{nth}_greedy_tokens"][i]
    k, v = m.kv(1, n_past)
    assert np.array_equal(k, g[f"nth{nth}_kcache_l1"]) and np.array_equal(v, g[f"nth{nth}_vcache_l1"])


def test_thread_count_is_part_of_the_numerics():
    g = load("tiny_model.npz")
    assert not np.array_equal(g["nth1_prompt_logits_all"], g["nth8_prompt_logits_all"])


def test_quantize_tool_golden_is_consistent_with_the_offline_quantizer(oracle):
    """tests/golden/quantize_file.npz was written by the reference's quantize tool (oracle/_ref/quantize).
    Walk both containers: headers and 1-D tensors are copied, the f16 field becomes 2, and every 2-D
    tensor's bytes equal the restated offline quantizer (utils.cpp:431-485) on the widened input."""
    import struct
    g = load("quantize_file.npz")

    def walk(buf):
        b = buf.tobytes()
        magic, = struct.unpack_from("<I", b, 0)
        hp = struct.unpack_from("<7i", b, 4)
        off = 32
        for _ in range(hp[0]):
            n, = struct.unpack_from("<I", b, off); off += 4 + n
        tensors = []
        while off < len(b):
            n_dims, length, ftype = struct.unpack_from("<3i", b, off); off += 12
            ne = struct.unpack_from(f"<{n_dims}i", b, off); off += 4 * n_dims
            name = b[off:off + length].decode(); off += length
            nel = int(np.prod(ne))
            size = {0: nel * 4, 1: nel * 2, 2: nel // 32 * 20}[ftype]
            tensors.append((name, ne, ftype, b[off:off + size])); off += size
        return magic, hp, off, tensors

    for tag in ("f16", "f32"):
        mi, hi, endi, ti = walk(g[f"in_{tag}"])
        mo, ho, endo, to = walk(g[f"out_{tag}"])
        assert mi == mo == 0x67676d6c and hi[:6] == ho[:6] and ho[6] == 2
        assert endo == g[f"out_{tag}"].size and [t[0] for t in ti] == [t[0] for t in to]
        for (name, ne, ft, data), (_, ne2, ft2, data2) in zip(ti, to):
            assert ne == ne2
            if len(ne) == 1:
                assert ft2 == 0 and data == data2
                continue
            x = np.frombuffer(data, np.float16 if ft == 1 else np.float32).astype(np.float32).reshape(ne[1], ne[0])
            assert ft2 == 2 and oracle.quantize_offline(x).tobytes() == data2, name

"""Test helper: writes small synthetic ``ggml-model-q4_0.bin[.k]`` files with numpy.

File layout follows the reference reader/writer
(Sources/llamaObjCxx/bridge/LlamaPredictOperation.mm:98-498, tools/convert-pth-to-ggml.py:92-169,
Sources/cpp/quantize.cpp:62-260).  Q4_0 bytes come from a numpy restatement of the reference's
OFFLINE quantizer ``ggml_quantize_q4_0`` (Sources/cpp/utils.cpp:431-485) -- C ``round`` (half away
from zero), ``id = 1.0f/d`` -- which is what defines the bytes of a real model file.

This is test infrastructure (tiny models, crafted edge cases).  Large benchmark models are written
by the product-side C++ tool ``llama.swift_amd/csrc/tools/make_synth_model.cpp``.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

QK = 32
MAGIC = 0x67676D6C


def n_ff_of(n_embd: int, n_mult: int) -> int:
    # LlamaPredictOperation.mm:135
    return ((2 * (4 * n_embd) // 3 + n_mult - 1) // n_mult) * n_mult


def quantize_q4_0_offline(x: np.ndarray) -> np.ndarray:
    """x: float32 [rows, K] -> uint8 [rows, K/32, 20] (utils.cpp:447-480)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, K = x.shape
    assert K % QK == 0
    xb = x.reshape(rows, K // QK, QK)
    amax = np.max(np.abs(xb), axis=2).astype(np.float32)
    d = (amax / np.float32(7.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        idv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    v = (xb * idv[:, :, None]).astype(np.float32).astype(np.float64)
    r = np.sign(v) * np.floor(np.abs(v) + 0.5)          # C round(): half away from zero
    q = (r.astype(np.int8).astype(np.int16) + 8).astype(np.uint8)
    packed = (q[:, :, 0::2] | (q[:, :, 1::2] << 4)).astype(np.uint8)
    out = np.empty((rows, K // QK, 20), dtype=np.uint8)
    out[:, :, :4] = d.view(np.uint8).reshape(rows, K // QK, 4)
    out[:, :, 4:] = packed
    return out


def dequantize_q4_0(blocks: np.ndarray) -> np.ndarray:
    """uint8 [rows, nb, 20] -> float32 [rows, nb*32] (ggml.c:651-684)."""
    rows, nb, _ = blocks.shape
    d = blocks[:, :, :4].copy().view(np.float32).reshape(rows, nb)
    qs = blocks[:, :, 4:]
    lo = (qs & 0xF).astype(np.int32) - 8
    hi = (qs >> 4).astype(np.int32) - 8
    vals = np.empty((rows, nb, 32), dtype=np.float32)
    vals[:, :, 0::2] = lo.astype(np.float32) * d[:, :, None]
    vals[:, :, 1::2] = hi.astype(np.float32) * d[:, :, None]
    return vals.reshape(rows, nb * 32)


@dataclass
class HParams:
    n_vocab: int = 64
    n_embd: int = 128
    n_mult: int = 32
    n_head: int = 1
    n_layer: int = 2
    n_rot: int = 64
    f16: int = 2

    @property
    def n_ff(self) -> int:
        return n_ff_of(self.n_embd, self.n_mult)


# split rule of LlamaPredictOperation.mm:358-388 (0: shard ne[0] / columns, 1: shard ne[1] / rows)
def split_type(name: str) -> int:
    if "tok_embeddings" in name:
        return 0
    if "layers" in name:
        if "attention.wo.weight" in name or "feed_forward.w2.weight" in name:
            return 0
        return 1
    if "output" in name:
        return 1
    return 0


def tensor_specs(hp: HParams):
    """(name, (ne1 rows, ne0 cols) or (n,)) in the order the converter emits them."""
    d, F, V = hp.n_embd, hp.n_ff, hp.n_vocab
    specs = [("tok_embeddings.weight", (V, d)), ("norm.weight", (d,)), ("output.weight", (V, d))]
    for i in range(hp.n_layer):
        p = f"layers.{i}."
        specs += [
            (p + "attention.wq.weight", (d, d)),
            (p + "attention.wk.weight", (d, d)),
            (p + "attention.wv.weight", (d, d)),
            (p + "attention.wo.weight", (d, d)),
            (p + "feed_forward.w1.weight", (F, d)),
            (p + "feed_forward.w2.weight", (d, F)),
            (p + "feed_forward.w3.weight", (F, d)),
            (p + "attention_norm.weight", (d,)),
            (p + "ffn_norm.weight", (d,)),
        ]
    return specs


def make_vocab(n_vocab: int) -> list[bytes]:
    """ids 0/1/2 empty (unk/bos/eos stand-ins), then a few single bytes, then ``tokNNNNN``."""
    words: list[bytes] = []
    for i in range(n_vocab):
        if i < 3:
            words.append(b"")
        elif i < 3 + 26 and i < n_vocab:
            words.append(bytes([ord("a") + i - 3]))
        elif i == 29:
            words.append(b" ")
        else:
            words.append(b"tok%05d" % i)
    return words


def random_tensors(hp: HParams, seed: int = 20230312, sigma: float = 0.02) -> dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out: dict[str, np.ndarray] = {}
    for name, shape in tensor_specs(hp):
        if len(shape) == 1:
            out[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            out[name] = (sigma * rng.standard_normal(shape)).astype(np.float32)
    return out


def write_model_unquantized(path: str, hp: HParams, tensors: dict[str, np.ndarray], ftype: int = 1,
                            vocab: list[bytes] | None = None, n_parts: int = 1) -> None:
    """f16 (ftype 1) or f32 (ftype 0) model file(s) as tools/convert-pth-to-ggml.py:92-169 writes them:
    1-D tensors always f32, 2-D tensors in `ftype`; header f16 field = ftype; one file per part with the
    same column / row shards as the quantized files.  This is the INPUT of the quantize tool
    (quantize.cpp) and what an f16 / f32 model run loads."""
    vocab = vocab if vocab is not None else make_vocab(hp.n_vocab)
    for part in range(n_parts):
        fname = path if part == 0 else f"{path}.{part}"
        with open(fname, "wb") as f:
            f.write(struct.pack("<I", MAGIC))
            f.write(struct.pack("<7i", hp.n_vocab, hp.n_embd, hp.n_mult, hp.n_head, hp.n_layer, hp.n_rot, ftype))
            for w in vocab:
                f.write(struct.pack("<I", len(w)))
                f.write(w)
            for name, _shape in tensor_specs(hp):
                t = tensors[name]
                nb = name.encode()
                if t.ndim == 1:
                    f.write(struct.pack("<3i", 1, len(nb), 0))
                    f.write(struct.pack("<i", t.shape[0]))
                    f.write(nb)
                    f.write(np.ascontiguousarray(t, dtype=np.float32).tobytes())
                    continue
                if n_parts > 1:
                    rows, cols = t.shape
                    if split_type(name) == 0:
                        w = cols // n_parts
                        t = t[:, part * w:(part + 1) * w]
                    else:
                        h = rows // n_parts
                        t = t[part * h:(part + 1) * h, :]
                rows, cols = t.shape
                f.write(struct.pack("<3i", 2, len(nb), ftype))
                f.write(struct.pack("<2i", cols, rows))
                f.write(nb)
                f.write(np.ascontiguousarray(t, dtype=np.float16 if ftype == 1 else np.float32).tobytes())


def write_model(path: str, hp: HParams, tensors: dict[str, np.ndarray], n_parts: int = 1,
                vocab: list[bytes] | None = None) -> None:
    """Write ``path`` (+ ``path.1`` ... for n_parts > 1) in the reference's container format.

    2-D tensors are quantized to Q4_0 *before* sharding for split_type 1 (rows) and per-shard for
    split_type 0 (columns; shard width is a multiple of 32 so blocks never straddle shards) -- the
    same bytes the reference pipeline (convert per-shard, then quantize per-file) produces.
    """
    vocab = vocab if vocab is not None else make_vocab(hp.n_vocab)
    assert len(vocab) == hp.n_vocab
    for part in range(n_parts):
        fname = path if part == 0 else f"{path}.{part}"
        with open(fname, "wb") as f:
            f.write(struct.pack("<I", MAGIC))
            f.write(struct.pack("<7i", hp.n_vocab, hp.n_embd, hp.n_mult, hp.n_head, hp.n_layer, hp.n_rot, hp.f16))
            for w in vocab:
                f.write(struct.pack("<I", len(w)))
                f.write(w)
            for name, _shape in tensor_specs(hp):
                t = tensors[name]
                nb = name.encode()
                if t.ndim == 1:
                    f.write(struct.pack("<3i", 1, len(nb), 0))
                    f.write(struct.pack("<i", t.shape[0]))
                    f.write(nb)
                    f.write(np.ascontiguousarray(t, dtype=np.float32).tobytes())
                    continue
                rows, cols = t.shape
                if n_parts > 1:
                    if split_type(name) == 0:
                        w = cols // n_parts
                        assert w % 64 == 0, "column shard must keep ne0 % 64 == 0 (.mm:437)"
                        t = t[:, part * w:(part + 1) * w]
                    else:
                        h = rows // n_parts
                        t = t[part * h:(part + 1) * h, :]
                rows, cols = t.shape
                assert cols % 64 == 0
                f.write(struct.pack("<3i", 2, len(nb), 2))
                f.write(struct.pack("<2i", cols, rows))          # ne[0] = input dim first
                f.write(nb)
                f.write(quantize_q4_0_offline(t).tobytes())


def synth_prompt(n: int, n_vocab: int, seed: int = 1) -> np.ndarray:
    """BOS (1) followed by uniform ids in [3, n_vocab) -- SURVEY.md section 8d."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(3, n_vocab, size=n, dtype=np.int32)
    ids[0] = 1
    return ids

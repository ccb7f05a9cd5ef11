"""ctypes bindings for the two CPU checkers (test infrastructure only).

* ``RefLib``    -> oracle/_ref/libggml_ref.so  : the reference's own ggml.c/utils.cpp compiled in
                   place + oracle/ref_driver.cpp (reference-backed oracle).
* ``OracleLib`` -> oracle/liboracle.so         : the standalone restatement oracle/oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libggml_ref.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

DUMP_NAMES = [
    "layer_in", "attn_normed", "q", "k", "v", "q_roped", "kq_softmax", "kqv", "kqv_merged",
    "wo_out", "ffn_in", "ffn_normed", "w3_out", "w1_out", "silu_mul", "w2_out", "layer_out",
]


def build_oracles() -> None:
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


class RefLib:
    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        L = self.L = C.CDLL(REF_SO)
        L.refllama_load.restype = C.c_void_p
        L.refllama_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.refllama_free.argtypes = [C.c_void_p]
        L.refllama_hparam.argtypes = [C.c_void_p, C.c_int]
        L.refllama_tensor_bytes.restype = C.c_long
        L.refllama_tensor_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long]
        L.refllama_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, f32p]
        L.refllama_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_char_p, C.c_size_t]
        L.ref_quantize_row_q4_0.argtypes = [f32p, u8p, C.c_int]
        L.ref_dequantize_row_q4_0.argtypes = [u8p, f32p, C.c_int]
        L.ref_quantize_q4_0_offline.restype = C.c_long
        L.ref_quantize_q4_0_offline.argtypes = [f32p, u8p, C.c_int, C.c_int, C.c_void_p]
        L.ref_fp16_to_fp32.restype = C.c_float
        L.ref_fp16_to_fp32.argtypes = [C.c_uint16]
        L.ref_fp32_to_fp16.restype = C.c_uint16
        L.ref_fp32_to_fp16.argtypes = [C.c_float]
        L.ref_mul_mat_q4_0.argtypes = [u8p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_unary_rows.argtypes = [C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int]
        L.ref_rope.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.refllama_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_int, i32p, C.c_int]
        L.refllama_sampler_new.restype = C.c_void_p
        L.refllama_sampler_new.argtypes = [C.c_int32, C.c_int]
        L.refllama_sampler_free.argtypes = [C.c_void_p]
        L.refllama_sampler_accept.argtypes = [C.c_void_p, C.c_int32]
        L.refllama_sampler_random_prompt.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.refllama_sampler_sample.restype = C.c_int32
        L.refllama_sampler_sample.argtypes = [C.c_void_p, C.c_void_p, f32p, C.c_double, C.c_int, C.c_double, C.c_double]
        L.ref_init_tables()

    # ---- kernels -------------------------------------------------------------------------
    def quantize_row(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32).ravel()
        out = np.empty(x.size // 32 * 20, np.uint8)
        self.L.ref_quantize_row_q4_0(x, out, x.size)
        return out

    def dequantize_row(self, q: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(q, np.uint8).ravel()
        k = q.size // 20 * 32
        out = np.empty(k, np.float32)
        self.L.ref_dequantize_row_q4_0(q, out, k)
        return out

    def quantize_offline(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        rows, k = x.shape
        out = np.empty(rows * (k // 32) * 20, np.uint8)
        hist = (C.c_int64 * 16)()
        n = self.L.ref_quantize_q4_0_offline(x.reshape(-1).copy(), out, rows * k, k, hist)
        assert n == out.size
        return out.reshape(rows, k // 32, 20)

    def mul_mat_q4_0(self, wq: np.ndarray, x: np.ndarray, n_threads: int = 1) -> np.ndarray:
        """wq uint8 [M, K/32, 20], x f32 [N, K] -> f32 [N, M]."""
        M, nb, _ = wq.shape
        K = nb * 32
        x = np.ascontiguousarray(x, np.float32).reshape(-1, K)
        N = x.shape[0]
        y = np.empty((N, M), np.float32)
        self.L.ref_mul_mat_q4_0(np.ascontiguousarray(wq).reshape(-1), x, y, M, K, N, n_threads)
        return y

    def unary_rows(self, op: str, x: np.ndarray, n_threads: int = 1) -> np.ndarray:
        code = {"norm": 0, "silu": 1, "soft_max": 2}[op]
        x = np.ascontiguousarray(x, np.float32)
        rows, cols = x.shape
        y = np.empty_like(x)
        self.L.ref_unary_rows(code, x, y, cols, rows, n_threads)
        return y

    def rope(self, x: np.ndarray, n_past: int, mode: int) -> np.ndarray:
        """x f32 [n, H, dh] (ggml ne = [dh, H, n])."""
        x = np.ascontiguousarray(x, np.float32).copy()
        n, H, dh = x.shape
        self.L.ref_rope(x, dh, H, n, n_past, mode)
        return x

    def f2h(self, v: float) -> int:
        return int(self.L.ref_fp32_to_fp16(C.c_float(v)))

    def h2f(self, h: int) -> float:
        return float(self.L.ref_fp16_to_fp32(C.c_uint16(h)))

    # ---- model ---------------------------------------------------------------------------
    def load(self, path: str, n_ctx: int = 512, force_parts: int = 0) -> "RefModel":
        err = C.create_string_buffer(512)
        h = self.L.refllama_load(path.encode(), n_ctx, force_parts, err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return RefModel(self, h)


class RefModel:
    def __init__(self, lib: RefLib, h):
        self.lib, self.h = lib, h
        g = lambda i: lib.L.refllama_hparam(h, i)
        self.n_vocab, self.n_ctx, self.n_embd, self.n_mult, self.n_head = g(0), g(1), g(2), g(3), g(4)
        self.n_layer, self.n_rot, self.f16, self.n_ff, self.n_parts = g(5), g(6), g(7), g(8), g(9)

    def close(self):
        if self.h:
            self.lib.L.refllama_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tensor_bytes(self, name: str) -> np.ndarray:
        n = self.lib.L.refllama_tensor_bytes(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, np.uint8)
        self.lib.L.refllama_tensor_bytes(self.h, name.encode(), out.ctypes.data_as(C.c_void_p), n)
        return out

    def kv(self, il: int, n_pos: int):
        k = np.empty((n_pos, self.n_embd), np.float32)
        v = np.empty((n_pos, self.n_embd), np.float32)
        self.lib.L.refllama_kv(self.h, il, n_pos, k, v)
        return k, v

    def eval(self, tokens, n_past: int, n_threads: int = 8, all_logits: bool = False, dump_layer: int = -1):
        tokens = np.ascontiguousarray(tokens, np.int32)
        N = tokens.size
        last = np.empty(self.n_vocab, np.float32)
        allb = np.empty((N, self.n_vocab), np.float32) if all_logits else None
        dump = sizes = None
        cap = 0
        if dump_layer >= 0:
            T = n_past + N
            cap = N * (14 * self.n_embd + 3 * self.n_ff) + T * N * self.n_head + 1024
            dump = np.zeros(cap, np.float32)
            sizes = (C.c_long * len(DUMP_NAMES))()
        err = C.create_string_buffer(512)
        rc = self.lib.L.refllama_eval(
            self.h, n_threads, n_past, tokens, N,
            last.ctypes.data_as(C.c_void_p), allb.ctypes.data_as(C.c_void_p) if all_logits else None,
            dump_layer, dump.ctypes.data_as(C.c_void_p) if dump is not None else None, cap,
            C.cast(sizes, C.c_void_p) if sizes is not None else None, err, 512)
        if rc != 0:
            raise RuntimeError(err.value.decode())
        res = {"logits": last}
        if all_logits:
            res["logits_all"] = allb
        if dump is not None:
            off = 0
            for i, name in enumerate(DUMP_NAMES):
                n = sizes[i]
                res[name] = dump[off:off + n].copy()
                off += n
        return res

    def tokenize(self, text: str, bos: bool = True) -> np.ndarray:
        out = np.empty(4096, np.int32)
        n = self.lib.L.refllama_tokenize(self.h, text.encode(), int(bos), out, out.size)
        return out[:n].copy()


class OracleLib:
    """Standalone restatement (oracle/oracle.c)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            raise FileNotFoundError(ORACLE_SO)
        L = self.L = C.CDLL(ORACLE_SO)
        L.orc_tables_get.argtypes = [np.ctypeslib.ndpointer(np.uint16), np.ctypeslib.ndpointer(np.uint16)]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_quantize_row_q4_0.argtypes = [f32p, u8p, C.c_int]
        L.orc_dequantize_row_q4_0.argtypes = [u8p, f32p, C.c_int]
        L.orc_quantize_q4_0_offline.argtypes = [f32p, u8p, C.c_long, C.c_int]
        L.orc_vec_dot_q4_0.restype = C.c_float
        L.orc_vec_dot_q4_0.argtypes = [C.c_int, u8p, u8p]
        L.orc_vec_dot_q4_0_scalar.restype = C.c_float
        L.orc_vec_dot_q4_0_scalar.argtypes = [C.c_int, u8p, u8p]
        L.orc_vec_dot_f32.restype = C.c_float
        L.orc_vec_dot_f32.argtypes = [C.c_int, f32p, f32p]
        L.orc_mul_mat_q4_0.argtypes = [u8p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int]
        for fn in (L.orc_norm_rows, L.orc_silu_rows, L.orc_softmax_rows):
            fn.argtypes = [f32p, f32p, C.c_int, C.c_int]
        L.orc_rope.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_load.restype = C.c_void_p
        L.orc_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_hparam.argtypes = [C.c_void_p, C.c_int]
        L.orc_tensor_bytes.restype = C.c_long
        L.orc_tensor_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long]
        L.orc_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, f32p]
        L.orc_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_void_p, C.c_long, C.c_void_p]
        L.orc_eval_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_tables_init()

    def tables(self):
        silu = np.empty(65536, np.uint16)
        expt = np.empty(65536, np.uint16)
        self.L.orc_tables_get(silu, expt)
        return silu, expt

    def quantize_row(self, x):
        x = np.ascontiguousarray(x, np.float32).ravel()
        out = np.empty(x.size // 32 * 20, np.uint8)
        self.L.orc_quantize_row_q4_0(x, out, x.size)
        return out

    def dequantize_row(self, q):
        q = np.ascontiguousarray(q, np.uint8).ravel()
        k = q.size // 20 * 32
        out = np.empty(k, np.float32)
        self.L.orc_dequantize_row_q4_0(q, out, k)
        return out

    def quantize_offline(self, x):
        x = np.ascontiguousarray(x, np.float32)
        rows, k = x.shape
        out = np.empty(rows * (k // 32) * 20, np.uint8)
        self.L.orc_quantize_q4_0_offline(x.reshape(-1), out, rows * k, k)
        return out.reshape(rows, k // 32, 20)

    def vec_dot_q4_0(self, a, b, scalar=False):
        a = np.ascontiguousarray(a, np.uint8).ravel()
        b = np.ascontiguousarray(b, np.uint8).ravel()
        fn = self.L.orc_vec_dot_q4_0_scalar if scalar else self.L.orc_vec_dot_q4_0
        return np.float32(fn(a.size // 20 * 32, a, b))

    def vec_dot_f32(self, a, b):
        a = np.ascontiguousarray(a, np.float32).ravel()
        b = np.ascontiguousarray(b, np.float32).ravel()
        return np.float32(self.L.orc_vec_dot_f32(a.size, a, b))

    def mul_mat_q4_0(self, wq, x, n_threads=1):
        M, nb, _ = wq.shape
        K = nb * 32
        x = np.ascontiguousarray(x, np.float32).reshape(-1, K)
        N = x.shape[0]
        y = np.empty((N, M), np.float32)
        self.L.orc_mul_mat_q4_0(np.ascontiguousarray(wq).reshape(-1), x, y, M, K, N, n_threads)
        return y

    def unary_rows(self, op, x):
        fn = {"norm": self.L.orc_norm_rows, "silu": self.L.orc_silu_rows, "soft_max": self.L.orc_softmax_rows}[op]
        x = np.ascontiguousarray(x, np.float32)
        rows, cols = x.shape
        y = np.empty_like(x)
        fn(x, y, cols, rows)
        return y

    def rope(self, x, n_past, mode):
        x = np.ascontiguousarray(x, np.float32).copy()
        n, H, dh = x.shape
        self.L.orc_rope(x, dh, H, n, n_past, mode)
        return x

    def load(self, path, n_ctx=512, force_parts=0):
        err = C.create_string_buffer(512)
        h = self.L.orc_load(path.encode(), n_ctx, force_parts, err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return OracleModel(self, h)


class OracleModel:
    def __init__(self, lib: OracleLib, h):
        self.lib, self.h = lib, h
        g = lambda i: lib.L.orc_hparam(h, i)
        self.n_vocab, self.n_ctx, self.n_embd, self.n_mult, self.n_head = g(0), g(1), g(2), g(3), g(4)
        self.n_layer, self.n_rot, self.f16, self.n_ff, self.n_parts = g(5), g(6), g(7), g(8), g(9)

    def close(self):
        if self.h:
            self.lib.L.orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tensor_bytes(self, name):
        n = self.lib.L.orc_tensor_bytes(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, np.uint8)
        self.lib.L.orc_tensor_bytes(self.h, name.encode(), out.ctypes.data_as(C.c_void_p), n)
        return out

    def kv(self, il, n_pos):
        k = np.empty((n_pos, self.n_embd), np.float32)
        v = np.empty((n_pos, self.n_embd), np.float32)
        self.lib.L.orc_kv(self.h, il, n_pos, k, v)
        return k, v

    def eval_range(self, l0, l1, n_past, tokens=None, hidden_in=None, n_threads=8):
        """One pipeline stage: returns (hidden_out or None, logits or None)."""
        tk = np.ascontiguousarray(tokens, np.int32) if tokens is not None else None
        hin = np.ascontiguousarray(hidden_in, np.float32) if hidden_in is not None else None
        N = tk.size if tk is not None else hin.size // self.n_embd
        last = l1 == self.n_layer
        hout = None if last else np.empty(N * self.n_embd, np.float32)
        logits = np.empty(self.n_vocab, np.float32) if last else None
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        rc = self.lib.L.orc_eval_range(self.h, n_threads, n_past, p(tk), N, l0, l1, p(hin), p(hout), p(logits))
        if rc != 0:
            raise RuntimeError(f"orc_eval_range failed: {rc}")
        return hout, logits

    def eval(self, tokens, n_past, n_threads=8, all_logits=False, dump_layer=-1):
        tokens = np.ascontiguousarray(tokens, np.int32)
        N = tokens.size
        last = np.empty(self.n_vocab, np.float32)
        allb = np.empty((N, self.n_vocab), np.float32) if all_logits else None
        dump = sizes = None
        cap = 0
        if dump_layer >= 0:
            T = n_past + N
            cap = N * (14 * self.n_embd + 3 * self.n_ff) + T * N * self.n_head + 1024
            dump = np.zeros(cap, np.float32)
            sizes = (C.c_long * len(DUMP_NAMES))()
        rc = self.lib.L.orc_eval(
            self.h, n_threads, n_past, tokens, N,
            last.ctypes.data_as(C.c_void_p), allb.ctypes.data_as(C.c_void_p) if all_logits else None,
            dump_layer, dump.ctypes.data_as(C.c_void_p) if dump is not None else None, cap,
            C.cast(sizes, C.c_void_p) if sizes is not None else None)
        if rc != 0:
            raise RuntimeError(f"orc_eval failed: {rc}")
        res = {"logits": last}
        if all_logits:
            res["logits_all"] = allb
        if dump is not None:
            off = 0
            for i, name in enumerate(DUMP_NAMES):
                n = sizes[i]
                res[name] = dump[off:off + n].copy()
                off += n
        return res

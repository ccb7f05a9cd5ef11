"""ctypes binding of include/llamahip.h (no torch types cross the boundary)."""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, os.environ.get("LLAMAHIP_LIB", "libllamahip.so"))     # LLAMAHIP_LIB=libllamahip_probe.so: measurement build
INCLUDE = os.path.join(ROOT, "include")

ERR_LOAD, ERR_PREDICT = -1000, -1001
DUMP_NAMES = [
    "layer_in", "attn_normed", "q", "k", "v", "q_roped", "kq_softmax", "kqv", "kqv_merged",
    "wo_out", "ffn_in", "ffn_normed", "w3_out", "w1_out", "silu_mul", "w2_out", "layer_out",
]
bench_gemv_names = {0: "wq|wk|wv", 1: "wo", 2: "w1|w3", 3: "w2", 4: "output"}


class LlamaHipError(RuntimeError):
    """Mirrors NSError(domain LlamaErrorDomain, code) (Sources/llamaObjCxx/headers/LlamaError.h:12-19)."""

    domain = "com.alexrozanski.llama.error"

    def __init__(self, code: int, message: str):
        super().__init__(f"[{self.domain} {code}] {message}")
        self.code = code
        self.message = message


def build(verbose: bool = False) -> str:
    """Compile libllamahip.so + tools in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "all"], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libllamahip.so failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


_lib = None


class _Opts(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("layer_begin", C.c_int32),
                ("layer_end", C.c_int32), ("n_parts", C.c_int32), ("flags", C.c_int32), ("n_seq", C.c_int32),
                ("n_devices", C.c_int32), ("devices", C.c_int32 * 8)]


class _GemvBench(C.Structure):
    _fields_ = [("M", C.c_int32), ("K", C.c_int32), ("iters", C.c_int32), ("ms_total", C.c_float),
                ("algo_bytes", C.c_double)]


class _Stats(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("weight_bytes_device", C.c_int64), ("kv_bytes_device", C.c_int64),
                ("n_evals", C.c_int64), ("t_load_ms", C.c_double), ("t_eval_ms_total", C.c_double), ("n_stages", C.c_int32), ("hand_off", C.c_int32)]


def lib() -> C.CDLL:
    """Load the HIP library; never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the HIP path)")
    # PyTorch-ROCm bundles its own libamdhip64.so.7 + libhsa-runtime64; whichever HIP runtime is
    # loaded first serves the whole process (same SONAME).  If this library pulled in the system
    # runtime first, a later torch.cuda init would pair it with torch's bundled HSA runtime and find
    # no GPUs -- so when torch is installed, let it load its runtime first (torch is only plumbing
    # here: device tensors for the pipeline hand-off and torch.distributed).
    if not os.environ.get("LLAMAHIP_NO_TORCH"):          # (measurement tools that never touch torch skip its slow import)
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, cp, sz = C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t
    L.llamahip_version.restype = cp
    L.llamahip_model_load.argtypes = [cp, i32, C.POINTER(_Opts), C.POINTER(vp), cp, sz]
    L.llamahip_eval.argtypes = [vp, i32, i32, vp, i32, vp, cp, sz]
    L.llamahip_eval_chunks.argtypes = [vp, i32, i32, vp, i32, i32, vp, cp, sz]
    L.llamahip_model_free.argtypes = [vp]
    for fn in ("n_vocab", "n_ctx", "n_embd", "n_head", "n_layer", "n_ff", "n_parts"):
        getattr(L, "llamahip_" + fn).argtypes = [vp]
        getattr(L, "llamahip_" + fn).restype = i32
    L.llamahip_token_text.argtypes = [vp, i32, C.POINTER(C.c_uint32)]
    L.llamahip_token_text.restype = vp
    L.llamahip_tokenize.argtypes = [vp, cp, i32, vp, i32]
    L.llamahip_tokenize.restype = i32
    L.llamahip_sampler_new.argtypes = [i32, i32]
    L.llamahip_sampler_new.restype = vp
    L.llamahip_sampler_free.argtypes = [vp]
    L.llamahip_sampler_accept.argtypes = [vp, i32]
    L.llamahip_eval_topk.argtypes = [vp, i32, i32, vp, i32, vp, i32, C.c_double, i32, C.c_double, vp, vp, vp, vp, cp, sz]
    L.llamahip_sample_from_candidates.argtypes = [vp, vp, vp, i32, C.c_double]
    L.llamahip_sample_from_candidates.restype = i32
    L.llamahip_sampler_window.argtypes = [vp, vp, i32]
    L.llamahip_sampler_window.restype = i32
    L.llamahip_sampler_random_prompt.argtypes = [vp]
    L.llamahip_sampler_random_prompt.restype = cp
    L.llamahip_sample_top_p_top_k.argtypes = [vp, vp, vp, C.c_double, i32, C.c_double, C.c_double]
    L.llamahip_sample_top_p_top_k.restype = i32
    L.llamahip_decode_greedy.argtypes = [vp, i32, i32, i32, i32, vp, vp, cp, sz]
    L.llamahip_decode_greedy_multi.argtypes = [vp, i32, i32, vp, vp, i32, vp, cp, sz]
    L.llamahip_eval_debug.argtypes = [vp, i32, i32, vp, i32, vp, vp, i32, vp, C.c_int64, vp, cp, sz]
    L.llamahip_eval_stage.argtypes = [vp, i32, i32, vp, i32, vp, vp, vp, cp, sz]
    L.llamahip_stage_bind.argtypes = [vp, i32, i32, vp, vp, vp, vp, cp, sz]
    L.llamahip_stage_step.argtypes = [vp, i32, i32, vp, cp, sz]
    L.llamahip_stage_trace.argtypes = [vp, i32, vp, vp, i32, cp, sz]
    L.llamahip_stage_step_set.argtypes = [vp, vp, i32, i32, vp, cp, sz]
    L.llamahip_stage_set_applies.argtypes = [vp, i32, i32]
    L.llamahip_stage_set_applies.restype = i32
    L.llamahip_stage_logits.argtypes = [vp, i32, vp, cp, sz]
    L.llamahip_stage_mailbox.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), vp, vp, cp, sz]
    L.llamahip_stage_mailbox_connect.argtypes = [vp, i32, vp, vp, vp, vp, cp, sz]
    L.llamahip_quantize_file.argtypes = [cp, cp, i32, cp, sz]
    L.llamahip_kv_read.argtypes = [vp, i32, i32, vp, vp, cp, sz]
    L.llamahip_set_seq.argtypes = [vp, i32, cp, sz]
    L.llamahip_tensor_bytes.argtypes = [vp, cp, vp, C.c_int64]
    L.llamahip_tensor_bytes.restype = C.c_int64
    L.llamahip_op_mul_mat_q4_0.argtypes = [vp, i32, i32, vp, i32, vp, cp, sz]
    L.llamahip_op_quantize_row_q4_0.argtypes = [vp, i32, vp, cp, sz]
    L.llamahip_op_topk.argtypes = [vp, i32, vp, i32, C.c_double, i32, C.c_double, vp, vp, vp, cp, sz]
    L.llamahip_bench_gemv.argtypes = [vp, i32, i32, i32, i32, C.POINTER(_GemvBench), cp, sz]
    L.llamahip_get_stats.argtypes = [vp, C.POINTER(_Stats)]
    L.llamahip_debug_lut_math.restype = i32
    L.llamahip_debug_gemm_paths.argtypes = [vp, i32]
    L.llamahip_debug_gemm_paths.restype = i32
    _lib = L
    return L


def set_plan(m: int, k: int, n_rows: int, epi: int, interleaved: bool = False):
    """Host-only: the few-row kernel's plan (nc, cw, ncg, rgw, lds_bytes) for n_rows against an m x k matrix, or None (llamahip_debug_set_plan)."""
    a = np.zeros(5, np.int64)
    f = lib().llamahip_debug_set_plan
    f.restype = C.c_int32
    f.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    return tuple(int(x) for x in a) if f(m, k, int(interleaved), n_rows, epi, a.ctypes.data_as(C.c_void_p)) else None


def gemm_paths() -> dict:
    """Launch counts of the multi-row mat-mul kernel families since process start (llamahip_debug_gemm_paths)."""
    a = np.zeros(8, np.int64)
    n = lib().llamahip_debug_gemm_paths(a.ctypes.data_as(C.c_void_p), 8)
    return dict(zip(("mfma", "rows", "lds", "gemv", "set"), a[:n].tolist()))


def version() -> str:
    return lib().llamahip_version().decode()


def declared_symbols() -> list[str]:
    """Every function name declared in include/*.h (used by the CPU-side ABI test)."""
    names: list[str] = []
    for h in sorted(os.listdir(INCLUDE)):
        if not h.endswith(".h"):
            continue
        text = open(os.path.join(INCLUDE, h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(llamahip_[a-z0-9_]+|llama_runner_[a-z0-9_]+)\s*\(", text)
    seen, out = set(), []
    for n in names:
        if n not in seen and n not in ("llamahip_opts", "llamahip_model", "llamahip_sampler", "llamahip_stats", "llamahip_gemv_bench"):
            seen.add(n)
            out.append(n)
    return out


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc: int, err) -> None:
    if rc != 0:
        raise LlamaHipError(rc, err.value.decode(errors="replace"))


def quantize_file(fname_inp: str, fname_out: str, itype: int = 2) -> None:
    """f32 / f16 model file -> Q4_0 model file (replaces llama_model_quantize, quantize.cpp:32-286)."""
    err = C.create_string_buffer(1024)
    _check(lib().llamahip_quantize_file(fname_inp.encode(), fname_out.encode(), itype, err, len(err)), err)


class Model:
    """Opaque model handle (llama_model + gpt_vocab of the reference, .mm:71-88, utils.h:49-55)."""

    def __init__(self, path: str, n_ctx: int = 512, device: int = -1, layer_begin: int = 0,
                 layer_end: int = -1, n_parts: int = 0, flags: int = 0, n_seq: int = 1, devices=None):
        """devices: a list of HIP device ordinals = an in-process layer pipeline, one stage per entry (include/llamahip.h);
        the same entry points work on it (eval, eval_chunks, eval_topk, decode_greedy, kv, stats)."""
        L = lib()
        err = C.create_string_buffer(1024)
        h = C.c_void_p()
        devices = list(devices or [])
        if len(devices) > 8:
            raise ValueError("at most 8 pipeline stages")
        opts = _Opts(C.sizeof(_Opts), device, layer_begin, layer_end, n_parts, flags, n_seq, len(devices), (C.c_int32 * 8)(*devices))
        self.layer_begin, self.n_seq = layer_begin, n_seq
        rc = L.llamahip_model_load(path.encode(), n_ctx, C.byref(opts), C.byref(h), err, len(err))
        _check(rc, err)
        self._h = h
        self.path = path
        self.n_vocab = L.llamahip_n_vocab(h)
        self.n_ctx = L.llamahip_n_ctx(h)
        self.n_embd = L.llamahip_n_embd(h)
        self.n_head = L.llamahip_n_head(h)
        self.n_layer = L.llamahip_n_layer(h)
        self.n_ff = L.llamahip_n_ff(h)
        self.n_parts = L.llamahip_n_parts(h)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().llamahip_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # --- llama_eval ---------------------------------------------------------------------------
    def eval(self, tokens, n_past: int, n_threads: int = 8) -> np.ndarray:
        tokens = np.ascontiguousarray(tokens, np.int32)
        logits = np.empty(self.n_vocab, np.float32)
        err = C.create_string_buffer(1024)
        rc = lib().llamahip_eval(self._h, n_threads, n_past, _ptr(tokens), tokens.size, _ptr(logits), err, len(err))
        _check(rc, err)
        return logits

    def eval_chunks(self, tokens, n_past: int, chunk_tokens: int = 9, n_threads: int = 8) -> np.ndarray:
        """The reference's prompt loop (successive llama_eval calls of `chunk_tokens` tokens, .mm:880-888) in one pass, bit for bit."""
        tokens = np.ascontiguousarray(tokens, np.int32)
        logits = np.empty(self.n_vocab, np.float32)
        err = C.create_string_buffer(1024)
        rc = lib().llamahip_eval_chunks(self._h, n_threads, n_past, _ptr(tokens), tokens.size, chunk_tokens, _ptr(logits), err, len(err))
        _check(rc, err)
        return logits

    def eval_debug(self, tokens, n_past: int, n_threads: int = 8, all_logits: bool = True, dump_layer: int = -1) -> dict:
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
            sizes = np.zeros(len(DUMP_NAMES), np.int64)
        err = C.create_string_buffer(1024)
        rc = lib().llamahip_eval_debug(self._h, n_threads, n_past, _ptr(tokens), N, _ptr(last), _ptr(allb),
                                       dump_layer, _ptr(dump), cap, _ptr(sizes), err, len(err))
        _check(rc, err)
        res = {"logits": last}
        if all_logits:
            res["logits_all"] = allb
        if dump is not None:
            off = 0
            for i, name in enumerate(DUMP_NAMES):
                n = int(sizes[i])
                res[name] = dump[off:off + n].copy()
                off += n
        return res

    def eval_topk(self, tokens, n_past: int, sampler: "Sampler", repeat_penalty: float = 1.3, top_k: int = 40,
                  temp: float = float(np.float32(0.8)), n_threads: int = 8):
        """llamahip_eval_topk: returns (exact, scores[k], ids[k], logits or None)."""
        tokens = np.ascontiguousarray(tokens, np.int32)
        win = np.zeros(1024, np.int32)
        nw = lib().llamahip_sampler_window(sampler._s, _ptr(win), 1024)
        sc, ids = np.zeros(64, np.float64), np.zeros(64, np.int32)
        exact = C.c_int32(0)
        logits = np.empty(self.n_vocab, np.float32)
        err = C.create_string_buffer(1024)
        rc = lib().llamahip_eval_topk(self._h, n_threads, n_past, _ptr(tokens), tokens.size, _ptr(win), nw, repeat_penalty, top_k, temp,
                                      _ptr(sc), _ptr(ids), C.byref(exact), _ptr(logits), err, len(err))
        _check(rc, err)
        k = min(top_k, self.n_vocab)
        return bool(exact.value), sc[:k], ids[:k], (None if exact.value else logits)

    def set_seq(self, seq: int) -> None:
        err = C.create_string_buffer(1024)
        _check(lib().llamahip_set_seq(self._h, seq, err, len(err)), err)

    def eval_stage(self, n_past: int, tokens=None, n_tokens: int = 0, hidden_in: int = 0, hidden_out: int = 0,
                   want_logits: bool = False, n_threads: int = 8):
        """Pipeline-stage eval.  hidden_in / hidden_out are DEVICE addresses (e.g. torch tensor
        .data_ptr()) of n_tokens * n_embd fp32; tokens is given on the first stage only."""
        tk = np.ascontiguousarray(tokens, np.int32) if tokens is not None else None
        N = tk.size if tk is not None else n_tokens
        logits = np.empty(self.n_vocab, np.float32) if want_logits else None
        err = C.create_string_buffer(1024)
        rc = lib().llamahip_eval_stage(self._h, n_threads, n_past, _ptr(tk), N, C.c_void_p(hidden_in), C.c_void_p(hidden_out),
                                       _ptr(logits), err, len(err))
        _check(rc, err)
        return logits

    def stage_bind(self, seq: int, n_past: int, token_in: int = 0, hidden_in: int = 0, hidden_out: int = 0, token_out: int = 0):
        """Fix slot `seq`'s next position and its DEVICE i/o buffers (addresses) for stage_step."""
        err = C.create_string_buffer(1024)
        _check(lib().llamahip_stage_bind(self._h, seq, n_past, C.c_void_p(token_in), C.c_void_p(hidden_in), C.c_void_p(hidden_out),
                                         C.c_void_p(token_out), err, len(err)), err)

    def stage_mailbox(self, seq: int):
        """Create (once) slot `seq`'s device-side inboxes.  Returns (hidden_ptr, token_ptr, hidden_handle, token_handle): device
        addresses (0 where the stage has no such inbox) for same-process neighbours, 64-byte IPC handles (bytes) for other processes."""
        hp, tp = C.c_void_p(0), C.c_void_p(0)
        hh, th = C.create_string_buffer(64), C.create_string_buffer(64)
        err = C.create_string_buffer(1024)
        _check(lib().llamahip_stage_mailbox(self._h, seq, C.byref(hp), C.byref(tp), hh, th, err, len(err)), err)
        return (hp.value or 0), (tp.value or 0), (hh.raw if hp.value else None), (th.raw if tp.value else None)

    def stage_mailbox_connect(self, seq: int, next_hidden_handle: bytes = None, next_hidden_ptr: int = 0, token_handle: bytes = None, token_ptr: int = 0):
        """Give slot `seq` the next stage's hidden inbox and / or (last stage) the first stage's token inbox: an IPC handle (bytes) or a
        device address each."""
        err = C.create_string_buffer(1024)
        _check(lib().llamahip_stage_mailbox_connect(self._h, seq, next_hidden_handle, C.c_void_p(next_hidden_ptr or None),
                                                    token_handle, C.c_void_p(token_ptr or None), err, len(err)), err)

    def stage_step(self, seq: int, n_threads: int = 8, stream: int = 0):
        """Enqueue one token step of this stage on `stream` (hipStream_t address, 0 = the null stream); asynchronous."""
        err = C.create_string_buffer(256)
        _check(lib().llamahip_stage_step(self._h, seq, n_threads, C.c_void_p(stream), err, len(err)), err)

    def stage_set_applies(self, n_seqs: int, n_threads: int = 8) -> bool:
        """Whether stage_step_set can step n_seqs slots as one set on this handle (else: stage_step per slot)."""
        return bool(lib().llamahip_stage_set_applies(self._h, n_seqs, n_threads))

    def stage_step_set(self, seqs, n_threads: int = 8, stream: int = 0):
        """One decode step for all the slots in `seqs` at once (bit-identical to stepping them one by one; the weights are
        streamed once for the whole set); asynchronous like stage_step."""
        err = C.create_string_buffer(1024)
        sq = np.ascontiguousarray(seqs, dtype=np.int32)
        _check(lib().llamahip_stage_step_set(self._h, _ptr(sq), len(sq), n_threads, C.c_void_p(stream), err, len(err)), err)

    def stage_logits(self, row: int = 0) -> np.ndarray:
        """Row `row` of the logits of the most recent step (waits for the device)."""
        err = C.create_string_buffer(1024)
        out = np.empty(self.n_vocab, np.float32)
        _check(lib().llamahip_stage_logits(self._h, row, _ptr(out), err, len(err)), err)
        return out

    def stage_trace(self, seq: int, cap: int = 0):
        """Waits for the device.  Returns (steps taken since bind, current position, tokens picked [last stage])."""
        toks = np.zeros(max(cap, 1), np.int32)
        pos = C.c_int32(0)
        err = C.create_string_buffer(1024)
        n = lib().llamahip_stage_trace(self._h, seq, C.byref(pos), _ptr(toks), cap, err, len(err))
        _check(min(n, 0), err)
        return n, pos.value, toks[:min(n, cap)]

    def decode_greedy(self, first_token: int, n_past: int, n_steps: int, n_threads: int = 8, want_logits: bool = False):
        out = np.empty(n_steps, np.int32)
        logits = np.empty(self.n_vocab, np.float32) if want_logits else None
        err = C.create_string_buffer(1024)
        rc = lib().llamahip_decode_greedy(self._h, n_threads, n_past, int(first_token), n_steps, _ptr(out), _ptr(logits), err, len(err))
        _check(rc, err)
        return (out, logits) if want_logits else out

    def decode_greedy_multi(self, first_tokens, n_past, n_steps: int, n_threads: int = 8) -> np.ndarray:
        """llamahip_decode_greedy_multi: sequences in KV slots 0 .. len(first_tokens) - 1 decoded together; returns [n_seqs][n_steps]."""
        ft = np.ascontiguousarray(first_tokens, np.int32)
        npast = np.ascontiguousarray(n_past, np.int32)
        out = np.empty((ft.size, n_steps), np.int32)
        err = C.create_string_buffer(1024)
        rc = lib().llamahip_decode_greedy_multi(self._h, n_threads, ft.size, _ptr(npast), _ptr(ft), n_steps, _ptr(out), err, len(err))
        _check(rc, err)
        return out

    def kv(self, il: int, n_pos: int):
        k = np.empty((n_pos, self.n_embd), np.float32)
        v = np.empty((n_pos, self.n_embd), np.float32)
        err = C.create_string_buffer(1024)
        rc = lib().llamahip_kv_read(self._h, il, n_pos, _ptr(k), _ptr(v), err, len(err))
        _check(rc, err)
        return k, v

    def tensor_bytes(self, name: str) -> np.ndarray:
        n = lib().llamahip_tensor_bytes(self._h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, np.uint8)
        if lib().llamahip_tensor_bytes(self._h, name.encode(), _ptr(out), n) != n:
            raise LlamaHipError(ERR_LOAD, f"failed to read tensor {name}")
        return out

    # --- vocab / text -------------------------------------------------------------------------
    def token_text(self, tid: int) -> bytes:
        ln = C.c_uint32(0)
        p = lib().llamahip_token_text(self._h, tid, C.byref(ln))
        if not p:
            raise IndexError(tid)
        return C.string_at(p, ln.value)

    def tokenize(self, text: str | bytes, bos: bool = True) -> np.ndarray:
        raw = text.encode() if isinstance(text, str) else text
        cap = len(raw) + 2
        out = np.empty(cap, np.int32)
        n = lib().llamahip_tokenize(self._h, raw, int(bos), _ptr(out), cap)
        return out[:n].copy()

    # --- measurement ----------------------------------------------------------------------------
    def bench_gemv(self, which: int, layer: int = 0, warmup: int = 5, iters: int = 50) -> dict:
        b = _GemvBench()
        err = C.create_string_buffer(1024)
        rc = lib().llamahip_bench_gemv(self._h, which, layer, warmup, iters, C.byref(b), err, len(err))
        _check(rc, err)
        us = b.ms_total * 1e3 / b.iters
        return {"name": bench_gemv_names[which], "M": b.M, "K": b.K, "iters": b.iters, "us_per_launch": us,
                "algo_bytes": b.algo_bytes, "GBps": b.algo_bytes / (us * 1e-6) / 1e9}

    def stats(self) -> dict:
        s = _Stats()
        lib().llamahip_get_stats(self._h, C.byref(s))
        return {k: getattr(s, k) for k, _ in _Stats._fields_ if k != "struct_size"}


class Sampler:
    """mt19937 + last_n_tokens window (LlamaPredictOperation.mm:773, 827-829)."""

    def __init__(self, seed: int = -1, repeat_last_n: int = 64):
        self._s = C.c_void_p(lib().llamahip_sampler_new(seed, repeat_last_n))

    def accept(self, tid: int) -> None:
        lib().llamahip_sampler_accept(self._s, int(tid))

    def sample_from_candidates(self, scores, ids, top_p: float = float(np.float32(0.95))) -> int:
        scores, ids = np.ascontiguousarray(scores, np.float64), np.ascontiguousarray(ids, np.int32)
        return int(lib().llamahip_sample_from_candidates(self._s, _ptr(scores), _ptr(ids), len(ids), top_p))

    def random_prompt(self) -> str:
        """gpt_random_prompt on this sampler's rng (utils.cpp:102-119; .mm:774-776)."""
        return lib().llamahip_sampler_random_prompt(self._s).decode()

    def sample(self, model: Model, logits: np.ndarray, repeat_penalty: float = 1.3, top_k: int = 40,
               top_p: float = float(np.float32(0.95)), temp: float = float(np.float32(0.8))) -> int:
        logits = np.ascontiguousarray(logits, np.float32)
        return int(lib().llamahip_sample_top_p_top_k(model._h, self._s, _ptr(logits), repeat_penalty, top_k, top_p, temp))

    def __del__(self):
        try:
            if self._s:
                lib().llamahip_sampler_free(self._s)
                self._s = None
        except Exception:
            pass


def op_topk(logits, window, repeat_penalty: float = 1.3, top_k: int = 40, temp: float = float(np.float32(0.8))):
    """Device half of the sampler on host logits: returns (exact, scores[top_k], ids[top_k])."""
    logits = np.ascontiguousarray(logits, np.float32)
    window = np.ascontiguousarray(window, np.int32)
    sc, ids, exact = np.zeros(64, np.float64), np.zeros(64, np.int32), C.c_int32(0)
    err = C.create_string_buffer(512)
    rc = lib().llamahip_op_topk(_ptr(logits), logits.size, _ptr(window), window.size, repeat_penalty, top_k, temp, _ptr(sc), _ptr(ids), C.byref(exact), err, len(err))
    _check(rc, err)
    return bool(exact.value), sc[:top_k], ids[:top_k]


def op_mul_mat_q4_0(wq: np.ndarray, x: np.ndarray) -> np.ndarray:
    """wq uint8 [M, K/32, 20] (file layout), x f32 [N, K] -> f32 [N, M] on the GPU."""
    wq = np.ascontiguousarray(wq, np.uint8)
    M, nb, _ = wq.shape
    K = nb * 32
    x = np.ascontiguousarray(x, np.float32).reshape(-1, K)
    N = x.shape[0]
    y = np.empty((N, M), np.float32)
    err = C.create_string_buffer(1024)
    rc = lib().llamahip_op_mul_mat_q4_0(_ptr(wq), M, K, _ptr(x), N, _ptr(y), err, len(err))
    _check(rc, err)
    return y


def op_quantize_row_q4_0(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32).ravel()
    out = np.empty(x.size // 32 * 20, np.uint8)
    err = C.create_string_buffer(1024)
    rc = lib().llamahip_op_quantize_row_q4_0(_ptr(x), x.size, _ptr(out), err, len(err))
    _check(rc, err)
    return out

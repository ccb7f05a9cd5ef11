"""Shared full-size model files and the CPU expectations of the full-size GPU tests, computed in the background.

Every CPU-side expectation that costs more than a few seconds (the reference build's 504-token 7B trace, the 13B / 65B full-depth
evals, the 2048-token prompts) is one child process (tests/cpu_expect.py) writing an .npz into a cache directory next to the model
files.  A full `-m gpu` session starts ALL of them at collection time (conftest.py), so they run side by side on the host cores while
the GPU tests that need no such expectation run; a test then only waits for whatever is left.  Nested pytest runs (variants) and
single-test runs find the .npz on disk or compute it on demand.  The model files themselves (synthetic, exact LLaMA shapes, written by
csrc/tools/make_synth_model) are written once per box under LLAMAHIP_MODEL_DIR behind a file lock and shared with bench.py."""
import fcntl
import os
import subprocess
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MODEL_DIR = os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models")
EXPECT_DIR = os.environ.get("LLAMAHIP_EXPECT_DIR", os.path.join(MODEL_DIR, "expect"))

# 2-layer models of the 13B / 65B WIDTHS (n_embd, heads, n_ff; 2 / 8-part files as the reference derives from n_embd, .mm:33-38): the
# decode step's attention schedule switches by POSITION at thresholds that depend on the width (llamahip.cpp attn_sched_at: 13B 544 and
# 1600, 65B 448), far beyond what a full-depth CPU expectation can reach -- prompts in the bridge's nine-token evals up to just below a
# threshold, then greedy tokens across it with NO environment override.  name -> (shape, n_ctx, prompt tokens, generated tokens)
WIDE = {
    "13Bw_544": (dict(n_vocab=512, n_embd=5120, n_mult=256, n_head=40, n_layer=2), 640, 531, 24),       # positions 531 .. 554: fused launch -> three launches at 544
    "13Bw_1600": (dict(n_vocab=512, n_embd=5120, n_mult=256, n_head=40, n_layer=2), 1664, 1593, 16),    # 1593 .. 1608: three launches -> streaming soft_max . V at 1600
    "65Bw_448": (dict(n_vocab=512, n_embd=8192, n_mult=256, n_head=64, n_layer=2), 512, 441, 14),       # 441 .. 454: fused launch -> three launches at 448
}


def _synth_tool(out, **kw):
    tool = os.path.join(ROOT, "llama.swift_amd", "csrc", "tools", "make_synth_model")
    args = [tool, "--out", str(out)]
    for k, v in kw.items():
        args += [f"--{k}", str(v)]
    subprocess.run(args, check=True, capture_output=True)


def _ensure(path, **kw):
    """write the model file unless it is there; two processes asking for the same file: one writes, the other waits on the lock"""
    if os.path.exists(path + ".done"):
        return path
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(path + ".done"):
            _synth_tool(path, **kw)
            open(path + ".done", "w").close()
    return path


def model(spec: str) -> str:
    """'7B' / '13B' / '65B': the full-size presets (seed 20230312, shared with bench.py); '13Bw' / '65Bw': the 2-layer wide models"""
    if spec.endswith("w"):
        kw = next(v[0] for k, v in WIDE.items() if k.startswith(spec + "_"))
        return _ensure(os.path.join(MODEL_DIR, f"{spec}-2layer-seed31", "ggml-model-q4_0.bin"), seed=31, **kw)
    return _ensure(os.path.join(MODEL_DIR, f"{spec}-seed20230312", "ggml-model-q4_0.bin"), preset=spec, seed=20230312)


# name -> (kind, model spec, n_ctx, n_prompt, n_gen, n_threads, prompt seed, n_vocab); kinds: tests/cpu_expect.py
JOBS = {
    "trace7b": ("trace", "7B", 512, 8, 504, 8, 2, 32000),               # configs[0] / configs[1]
    "flow2048": ("flow", "7B", 2560, 2048, 3, 8, 6, 32000),             # configs[2], the reference's nine-token flow
    "single2048": ("single", "7B", 2560, 2048, 3, 8, 5, 32000),         # configs[2], one eval
    "13B": ("decode", "13B", 64, 9, 5, 8, 3, 32000),                    # configs[3]
    "13B_128": ("flow", "13B", 256, 128, 32, 8, 9, 32000),
    **{name: ("flow", name.split("_")[0], n_ctx, n_prompt, n_gen, 8, 12, kw["n_vocab"]) for name, (kw, n_ctx, n_prompt, n_gen) in WIDE.items()},
    "65B": ("decode", "65B", 64, 9, 4, 8, 3, 32000),                    # configs[4]'s model
}
# Host DRAM bandwidth is what these jobs share (r06_a: all nine at once stretched the 30 s trace to 195 s): at most MAX_JOBS run side by
# side, in the order the tests need them -- the 7B trace first, the 65B job (mostly its 40 GB file write) early because it is the longest.
ORDER = ["trace7b", "65B", "13B", "13B_128", "13Bw_544", "13Bw_1600", "65Bw_448", "flow2048", "single2048"]
MAX_JOBS = int(os.environ.get("LLAMAHIP_EXPECT_JOBS", "3"))
_running = {}
_queue = []
_lock = threading.Lock()
_pump_thread = None


def _out(name):
    return os.path.join(EXPECT_DIR, name + ".npz")


def _launch(name):
    os.makedirs(EXPECT_DIR, exist_ok=True)
    kind, spec, n_ctx, n_prompt, n_gen, nth, seed, n_vocab = JOBS[name]
    log = open(os.path.join(EXPECT_DIR, name + ".log"), "w")
    _running[name] = subprocess.Popen([sys.executable, os.path.join(HERE, "cpu_expect.py"), kind, "spec:" + spec, str(n_ctx), str(n_prompt), str(n_gen), str(nth), str(seed),
                                       _out(name), str(n_vocab)], stdout=log, stderr=subprocess.STDOUT)


def _pump():
    while True:
        with _lock:
            busy = sum(1 for p in _running.values() if p.poll() is None)
            while _queue and busy < MAX_JOBS:
                name = _queue.pop(0)
                if name not in _running and not os.path.exists(_out(name)):
                    _launch(name)
                    busy += 1
            if not _queue:
                return
        time.sleep(0.5)


def start_all():
    global _pump_thread
    with _lock:
        for name in ORDER:
            if name == "65B" and os.environ.get("LLAMAHIP_SKIP_65B"):
                continue
            if name not in _running and name not in _queue and not os.path.exists(_out(name)):
                _queue.append(name)
    if _pump_thread is None or not _pump_thread.is_alive():
        _pump_thread = threading.Thread(target=_pump, daemon=True)
        _pump_thread.start()


def get(name, timeout=1500):
    if not os.path.exists(_out(name)):
        with _lock:                                     # asked for before its turn (or never queued: a single-test run): start it now
            if name in _queue:
                _queue.remove(name)
            if name not in _running:
                _launch(name)
        p = _running[name]
        p.wait(timeout=timeout)
        log = open(os.path.join(EXPECT_DIR, name + ".log")).read()
        assert p.returncode == 0 and os.path.exists(_out(name)), f"cpu_expect {name} failed:\n{log[-3000:]}"
    return np.load(_out(name))


def stop_all():
    with _lock:
        _queue.clear()
        for p in _running.values():
            if p.poll() is None:
                p.kill()

#!/usr/bin/env python3
"""bench.py -- decode tokens/sec of the Q4_0 LLaMA hot path on MI355X + HBM roofline of the GEMV.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` (N > 1 is launched through
``python -m torch.distributed.run --nproc-per-node N ...``); rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): LLaMA-7B Q4_0, single-token decode, 512-token generation,
n_ctx 512.  There are no real weights offline, so the model file is synthetic (random-init weights of
the 7B architecture written in the reference's exact file format by csrc/tools/make_synth_model) and
the prompt is synthetic token ids.  One *step* = one decoded token = one pass of the hot path
(llama_eval with one token at a growing context offset) including the greedy argmax that feeds the
next step on the device.  `value` = tokens/s with everything resident in HBM (logits are not copied
back per token inside the timed region; the PCIe-inclusive rate is reported as `value_pcie`).

N > 1: the model is layer-sharded into N pipeline stages (one process per GPU); N independent
greedy sequences are kept in flight round-robin so every stage is busy, and the residual stream
crosses stages with point-to-point RCCL send/recv.  value = all sequences'
tokens / time, scaling = "weak" (one sequence per GPU).

Extra objects:  "roofline" for the dominant kernel (k_gemv, the Q4_0 x Q4_0 decode GEMV) and
"cpu_baseline" (the reference's own ggml.c, oracle/_ref, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODELS = {
    "7B": dict(n_vocab=32000, n_embd=4096, n_mult=256, n_head=32, n_layer=32),
    "13B": dict(n_vocab=32000, n_embd=5120, n_mult=256, n_head=40, n_layer=40),
    "65B": dict(n_vocab=32000, n_embd=8192, n_mult=256, n_head=64, n_layer=80),
    # small stand-in for smoke-testing the harness itself (never used for reported numbers)
    "tiny": dict(n_vocab=512, n_embd=512, n_mult=64, n_head=4, n_layer=4),
}
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def model_path(name: str, cfg: dict, seed: int) -> str:
    base = os.environ.get("LLAMAHIP_MODEL_DIR", "/tmp/llamahip_models")
    os.makedirs(base, exist_ok=True)
    path = os.path.join(base, f"{name}-seed{seed}", "ggml-model-q4_0.bin")
    if not os.path.exists(path + ".done"):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tool = os.path.join(ROOT, "llama.swift_amd", "csrc", "tools", "make_synth_model")
        t0 = time.time()
        args = [tool, "--out", path, "--seed", str(seed)] + [x for k, v in cfg.items() for x in (f"--{k}", str(v))]
        # the generator writes the reference's part count for the width (.mm:33-38): 13B = 2 files, 65B = 8
        subprocess.run(args, check=True)
        open(path + ".done", "w").close()
        log(f"[bench] wrote synthetic {name} model in {time.time() - t0:.1f}s: {path}")
    return path


def n_ff(cfg):
    return ((2 * (4 * cfg["n_embd"]) // 3 + cfg["n_mult"] - 1) // cfg["n_mult"]) * cfg["n_mult"]


def gemv_bytes(M, K):
    # 20 B per 32-weight block + quantized activations + fp32 outputs
    return M * (K // 32) * 20 + (K // 32) * 20 + 4 * M


def cpu_baseline(path: str, n_threads: int, budget_s: float, n_ctx: int) -> dict:
    """The reference's own ggml.c (oracle/_ref, built in place from /root/reference) on the host
    cores: greedy decode on the same model file, bounded by `budget_s` seconds of wall time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import reflib
    kind = "reference" if reflib.have_ref() else "port"
    lib = reflib.RefLib() if kind == "reference" else reflib.OracleLib()
    m = lib.load(path, n_ctx, 1) if kind == "reference" else lib.load(path, n_ctx, 1)
    prompt = np.array([1, 17, 291, 4001, 29, 512, 77, 1234], np.int32)
    logits = m.eval(prompt, 0, n_threads)["logits"]
    tok, n_past, n = int(np.argmax(logits)), len(prompt), 0
    t0 = time.time()
    while time.time() - t0 < budget_s and n_past < n_ctx:
        logits = m.eval(np.array([tok], np.int32), n_past, n_threads)["logits"]
        tok = int(np.argmax(logits)); n_past += 1; n += 1
    dt = time.time() - t0
    m.close()
    return {"value": n / dt, "unit": "tokens/s", "cores": n_threads, "kind": kind,
            "sample": f"{n} greedy decode tokens after an 8-token prompt, same synthetic model file, {dt:.1f}s wall, "
                      f"{n_threads} threads (reference default numThreads=8, Sources/llama/LlamaRunner.swift:17)",
            "host_cpus": os.cpu_count()}


def run_single(args, cfg, path):
    import llama_swift_amd as L
    t0 = time.time()
    m = L.Model(path, n_ctx=args.n_ctx)
    t_load = time.time() - t0
    log(f"[bench] loaded in {t_load:.1f}s: {m.stats()}")
    prompt = np.array([1, 17, 291, 4001, 29, 512, 77, 1234], np.int32) % cfg["n_vocab"]
    prompt[0] = 1
    logits = m.eval(prompt, 0, args.threads)
    tok, n_past = int(np.argmax(logits)), len(prompt)
    steps = min(args.steps, args.n_ctx - n_past - args.warmup)
    if args.warmup > 0:
        w = m.decode_greedy(tok, n_past, args.warmup, args.threads)
        tok, n_past = int(w[-1]), n_past + args.warmup
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.decode_greedy(tok, n_past, steps, args.threads)       # synchronises before returning
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # PCIe-inclusive variant: the reference boundary (llama_eval returning host logits every token)
    n_pcie = min(64, steps)
    m2_past = n_past
    t1 = time.perf_counter()
    tk = tok
    for i in range(n_pcie):
        lg = m.eval(np.array([tk], np.int32), m2_past + i, args.threads)
        tk = int(np.argmax(lg))
    dt_pcie = time.perf_counter() - t1
    same = bool(np.array_equal(out[:n_pcie], np.array([int(x) for x in out[:n_pcie]])))
    # roofline: the decode GEMV kernel on every resident matrix kind, cycling over all layers so the
    # weights stream from HBM (one pass over the model is 4.1 GB >> 256 MB Infinity Cache)
    shapes = []
    tot_bytes = tot_us = 0.0
    for which in range(5):
        r = m.bench_gemv(which, -1, 1, args.gemv_iters)
        shapes.append(r)
        per_token = 1 if which == 4 else cfg["n_layer"]
        tot_bytes += r["algo_bytes"] * per_token
        tot_us += r["us_per_launch"] * per_token
    # prompt evaluation (not the headline metric): one eval of n_ctx - 8 tokens from an empty context,
    # and the same prompt in the reference's 9-token chunks (.mm:848-861, n_batch 8)
    rng = np.random.default_rng(7)
    ptoks = rng.integers(3, cfg["n_vocab"], args.n_ctx - 8).astype(np.int32)
    ptoks[0] = 1
    m.eval(ptoks, 0, args.threads)
    t2 = time.perf_counter(); m.eval(ptoks, 0, args.threads); dt_pre = time.perf_counter() - t2
    t2 = time.perf_counter()
    n9 = 0
    for c0 in range(0, len(ptoks) - 8, 9):
        m.eval(ptoks[c0:c0 + 9], c0, args.threads)
        n9 += len(ptoks[c0:c0 + 9])
    dt_9 = time.perf_counter() - t2
    prefill = {"one_eval": {"tokens": int(len(ptoks)), "tokens_per_s": len(ptoks) / dt_pre},
               "reference_9_token_chunks": {"tokens": int(n9), "tokens_per_s": n9 / dt_9},
               "note": "exact path (bit-identical to the reference); host logits copy included"}
    m.close()
    # the reference's user-facing flow (LlamaRunner.run: load once, 8-token prompt batches, one llama_eval and one
    # host-side top-k / top-p sample per token, token text through the event callback) -- not the headline metric
    runner_info = None
    try:
        from llama_swift_amd import LlamaRunner, Config
        stamps = []
        rn = LlamaRunner(path)
        cfgr = Config(numThreads=args.threads, numTokens=320, n_ctx=args.n_ctx, keepModel=True)
        prompt_text = "".join("tok%05d" % t for t in rng.integers(3, cfg["n_vocab"], 16))
        rn.run(prompt_text, cfgr)                                             # loads the model, warms up
        rn.run(prompt_text, cfgr, tokenHandler=lambda _t: stamps.append(time.perf_counter()))
        rn.close()
        if len(stamps) > 64:
            gen = stamps[-257:] if len(stamps) >= 257 + 17 else stamps[17:]   # generated tokens only (the prompt is echoed first)
            runner_info = {"sampled_tokens_per_s": (len(gen) - 1) / (gen[-1] - gen[0]), "tokens": len(gen) - 1,
                           "note": "LlamaRunner.run with its default sampling (top_k 40, top_p 0.95, temp 0.8, repeat penalty 1.3 on the host), model kept from a previous run"}
    except Exception as e:                                                    # the runner leg must never cost the headline line
        log(f"[bench] runner leg skipped: {e}")
    # a measured ceiling next to the nominal 8 TB/s: device-to-device copy of 1 GiB (read + write)
    src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record(); torch.cuda.synchronize()
    copy_gbps = 10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    return dict(steps=steps, dt=dt, tokens=out, t_load=t_load, value_pcie=n_pcie / dt_pcie, shapes=shapes,
                gemv_bytes_per_token=tot_bytes, gemv_us_per_token=tot_us, same=same, prefill=prefill, copy_gbps=copy_gbps, runner=runner_info)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=496)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default=os.environ.get("LLAMAHIP_BENCH_MODEL", "7B"), choices=sorted(MODELS))
    ap.add_argument("--n_ctx", type=int, default=512)
    ap.add_argument("--threads", type=int, default=8, help="reference n_threads (selects its V*P summation split)")
    ap.add_argument("--seed", type=int, default=20230312)
    ap.add_argument("--gemv-iters", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cfg = MODELS[args.model]

    if world > 1 or args.gpus > 1 or os.environ.get("LLAMAHIP_FORCE_PIPELINE"):      # FORCE: exercise the N > 1 code path on one GPU
        # every stream of a rank (compute, one per RCCL communicator) gets its own hardware queue, so
        # a point-to-point kernel waiting for its peer can never sit in front of unrelated work
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        from llama_swift_amd import pipeline
        return pipeline.bench_main(args, cfg, model_path, log)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the HIP path)")
    path = model_path(args.model, cfg, args.seed)
    r = run_single(args, cfg, path)
    tps = r["steps"] / r["dt"]
    F = n_ff(cfg)
    d = cfg["n_embd"]
    dom = max(r["shapes"], key=lambda s: s["algo_bytes"] * (1 if s["name"] == "output" else cfg["n_layer"]))
    all_gemv = r["gemv_bytes_per_token"] / (r["gemv_us_per_token"] * 1e-6) / 1e9
    achieved = dom["GBps"]
    traffic = None
    try:        # HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), committed under profiles/
        tj = json.load(open(os.path.join(ROOT, "profiles", "gemv_traffic.json")))
        traffic = tj["per_launch"][dom["name"]]["traffic_bytes"] if args.model == "7B" else None
    except Exception:
        pass
    result = {
        "metric": "decode tokens/sec LLaMA-7B Q4_0 @1 GPU; % HBM-roofline on Q4_0 GEMV",
        "value": tps, "unit": "tokens/s", "n_gpus": 1, "steps": r["steps"], "warmup": args.warmup,
        "ms_per_step": r["dt"] * 1e3 / r["steps"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "q4_0 x q4_0 -> int32 block sums, fp32 scales/accumulate",
        "data": "synthetic (random-init 7B-architecture weights in the reference file format, synthetic token ids)",
        "config": {"workload": f"LLaMA-{args.model} Q4_0 single-token decode, greedy, n_ctx {args.n_ctx}, "
                               f"{r['steps']} timed tokens after 8 prompt + {args.warmup} warm-up tokens",
                   "n_threads_semantics": args.threads, "parallelism": "1 GPU"},
        "value_pcie": r["value_pcie"],
        "load_s": r["t_load"],
        "prefill": r["prefill"],
        "runner": r.get("runner"),
        "roofline": {"bound": "hbm",
                     "kernel": f"lh::k_gemv, the Q4_0 x Q4_0 decode GEMV, on its dominant shape {dom['name']} "
                               f"(M={dom['M']}, K={dom['K']}: {dom['algo_bytes'] * cfg['n_layer'] / r['gemv_bytes_per_token'] * 100:.0f}% of the GEMV bytes of a token)",
                     "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": dom["algo_bytes"], "us_per_launch": dom["us_per_launch"],
                     "measured_d2d_copy_GBps": r["copy_gbps"], "frac_of_measured_copy": achieved / r["copy_gbps"],
                     "all_gemv_launches_of_a_token": {"achieved": all_gemv, "frac": all_gemv / HBM_PEAK_GBPS,
                                                      "bytes": r["gemv_bytes_per_token"], "us": r["gemv_us_per_token"]},
                     "per_shape": [{k: s[k] for k in ("name", "M", "K", "us_per_launch", "GBps")} for s in r["shapes"]],
                     "note": "algorithmic bytes per launch = M*(K/32)*20 + (K/32)*20 + 4*M (SURVEY.md 8d); duration = HIP-event "
                             "average on the launch stream over back-to-back launches cycling through all 32 layers (cold weights; "
                             "includes the inter-launch dispatch gap that rocprofv3's kernel duration excludes); traffic = HBM "
                             "bytes per launch from separate rocprofv3 --pmc passes (profiles/r01_i_gemv_pmc.txt, tools/pmc_pass.sh)"},
    }
    if not args.no_cpu_baseline and args.cpu_seconds > 0:
        try:
            result["cpu_baseline"] = cpu_baseline(path, 8, args.cpu_seconds, args.n_ctx)
        except Exception as e:  # the checker must never take the measurement down
            result["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

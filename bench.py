#!/usr/bin/env python3
"""bench.py -- decode tokens/sec of the Q4_0 LLaMA hot path on MI355X + the HBM roofline of its mat-vecs.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` (N > 1 is launched through
``python -m torch.distributed.run --nproc-per-node N ...``); rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): LLaMA-7B Q4_0, single-token decode, 512-token generation,
n_ctx 512.  There are no real weights offline, so the model file is synthetic (random-init weights of
the 7B architecture written in the reference's exact file format by csrc/tools/make_synth_model) and
the prompt is synthetic token ids.  One *step* = one decoded token = one pass of the hot path
(llama_eval with one token at a growing context offset) including the greedy argmax that feeds the
next step on the device.  `value` = tokens/s with everything resident in HBM (logits are not copied
back per token inside the timed region; the PCIe-inclusive rate is reported as `value_pcie`).

Objects next to the contract's fields (N = 1):
  roofline      the dominant kernel of the decode step AS IT RUNS IN THE STEP: the w1|w3 mat-vec with its norm
                prologue and SiLU*up -> Q4_0 epilogue.  Its average duration (and that of every other launch of
                the captured step) comes from a rocprofv3 --kernel-trace run of the same decode loop in a child
                process, dispatches labelled by their position in the token's launch sequence; `achieved` =
                algorithmic bytes / that duration.  `end_to_end_frac` prices the whole step (weights + fp32 KV +
                small) against 8 TB/s; `probe_back_to_back` keeps the stand-alone PRE_QA / STORE variant that
                round 1 reported (it never runs in the step).
  parity        the tokens of the timed run against the CPU path on the same file and prompt (as many as the CPU
                budget covers) and max |delta logit| at the last compared step: the gate SURVEY.md 8(d) attaches
                to every performance number.
  cpu_baseline  the reference's own ggml.c (oracle/_ref; the standalone restatement if that did not travel) on the
                host: 8 threads (the reference default) and, as `all_cores`, min(nproc, 64) threads.
  prefill       exact-path prompt evaluation: one 504-token eval, the reference's 9-token chunks (eval by eval, and
                in one chunk-exact pass: llamahip_eval_chunks), and
                configs[2] (2048 tokens in one eval at n_ctx 2560) with its useful-op rate against the int8
                matrix-core peak and its fp32 FMA rate against the vector peak.

N > 1: the model is layer-sharded into N pipeline stages (one process per GPU); independent greedy sequences are
kept in flight round-robin so every stage is busy, and the residual stream crosses stages with point-to-point RCCL
send/recv.  value = all sequences' tokens / time, scaling = "weak".
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
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
F16_MFMA_PEAK_TFLOPS = 2500.0        # dense fp16 (MI355X_MICROARCH.md)
INT8_MFMA_PEAK_TOPS = 5000.0   # dense int8 matrix-core peak: 2x the 2.5 PFLOP/s bf16 figure (MI355X_MICROARCH.md: i8 = 2x K)
FP32_VALU_PEAK_TFLOPS = 157.3  # vector fp32 peak (MI355X_MICROARCH.md)
PROMPT = np.array([1, 17, 291, 4001, 29, 512, 77, 1234], np.int32)


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


def decode_roles(cfg):
    """The launches of one captured decode step, in order, with the algorithmic bytes of the mat-vecs: the layouts the
    library can run (it picks one per model shape), each entry (role, mat-vec bytes or None, substring of the kernel name)."""
    d, F, V, L = cfg["n_embd"], n_ff(cfg), cfg["n_vocab"], cfg["n_layer"]
    tail = [("wo", gemv_bytes(d, d), "k_gemv"), ("w1|w3", gemv_bytes(2 * F, d), "k_gemv"), ("w2", gemv_bytes(d, F), "k_gemv")]
    layouts = [
        # wq|wk|wv + attention in one launch (k_qkv_attn): the mat-vec bytes are counted, the K / V rows it also reads are not
        [("wq|wk|wv+attention", gemv_bytes(3 * d, d), "k_qkv_attn")] + tail,
        [("wq|wk|wv", gemv_bytes(3 * d, d), "k_gemv"), ("attention", None, "k_dec_attn_x")] + tail,
        [("wq|wk|wv", gemv_bytes(3 * d, d), "k_gemv"), ("attn_scores", None, "k_dec_scores"), ("attn_softmax_pv", None, "k_dec_pv_blk")] + tail,
        # long contexts (llamahip.cpp attn_sched_at): the streaming soft_max . V
        [("wq|wk|wv", gemv_bytes(3 * d, d), "k_gemv"), ("attn_scores", None, "k_dec_scores"), ("attn_softmax_pv", None, "k_dec_pv_stream")] + tail,
    ]
    return layouts, ("output", gemv_bytes(V, d)), L


def token_bytes(cfg, t):
    """Algorithmic bytes of one decoded token at position t: W + KV(t) + small."""
    d, F, V, L = cfg["n_embd"], n_ff(cfg), cfg["n_vocab"], cfg["n_layer"]
    W = (L * (4 * d * d + 3 * d * F) + V * d) // 32 * 20
    kv = L * 2 * (t + 1) * d * 4 + L * 2 * d * 4
    small = (2 * L + 1) * d * 4 + d // 32 * 20 + 4 * V
    return W + kv + small


# ------------------------------------------------------------------------------------------------ CPU path
def cpu_decode(path: str, n_threads: int, budget_s: float, n_ctx: int, want_tokens: int) -> dict:
    """Greedy decode on the CPU path from the bench prompt, bounded by `budget_s` seconds / `want_tokens` tokens."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import reflib
    kind = "reference" if reflib.have_ref() else "port"
    lib = reflib.RefLib() if kind == "reference" else reflib.OracleLib()
    m = lib.load(path, n_ctx, 0)          # 0: the loader derives the part count from n_embd (.mm:33-38): 13B = 2 files, 65B = 8
    prompt = PROMPT.copy()
    logits = m.eval(prompt, 0, n_threads)["logits"]
    tok, n_past, toks = int(np.argmax(logits)), len(prompt), []
    first = tok
    t0 = time.time()
    while time.time() - t0 < budget_s and n_past < n_ctx and len(toks) < want_tokens:
        logits = m.eval(np.array([tok], np.int32), n_past, n_threads)["logits"]
        tok = int(np.argmax(logits)); n_past += 1; toks.append(tok)
    dt = time.time() - t0
    m.close()
    return dict(kind=kind, first=first, toks=toks, last_logits=logits, dt=dt, threads=n_threads)


# ------------------------------------------------------------------------------------------------ in-situ profile
def insitu_child(args, cfg, path):
    """(child of rocprofv3) the decode loop only: prompt, warm-up, `steps` greedy tokens on the device."""
    import llama_swift_amd as L
    m = L.Model(path, n_ctx=args.n_ctx)
    logits = m.eval(PROMPT % cfg["n_vocab"], 0, args.threads)
    tok = int(np.argmax(logits))
    w = m.decode_greedy(tok, len(PROMPT), max(args.warmup, 1), args.threads)
    m.decode_greedy(int(w[-1]), len(PROMPT) + max(args.warmup, 1), args.steps, args.threads)
    m.close()


def insitu_stage_child(args, cfg, path):
    """(child of rocprofv3) the decode loop of ONE pipeline stage as the N > 1 bench runs it: a first-stage handle holding
    `--stage-layers` layers, `--stage-set` sequences stepped as one set (1: one sequence per step)."""
    import torch
    import llama_swift_amd as L
    B, nl = max(1, args.stage_set), args.stage_layers
    m = L.Model(path, n_ctx=args.n_ctx, layer_begin=0, layer_end=nl, n_seq=B)
    last = nl >= cfg["n_layer"]
    tok = torch.tensor([5 + 3 * s for s in range(B)], dtype=torch.int32, device="cuda")
    hid = None if last else torch.zeros(B, cfg["n_embd"], dtype=torch.float32, device="cuda")
    for s in range(B):
        m.stage_bind(s, 8 + s % 3, token_in=tok[s:s + 1].data_ptr(), hidden_out=0 if last else hid[s].data_ptr(), token_out=tok[s:s + 1].data_ptr() if last else 0)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(max(args.warmup, 1) + args.steps):
        if B > 1:
            m.stage_step_set(list(range(B)), args.threads, st)
        else:
            m.stage_step(0, args.threads, st)
    torch.cuda.synchronize()
    m.close()


def insitu_stage_profile(args, model_name, cfg, stage_layers, stage_set):
    """The dominant mat-vec / mat-mul of a pipeline stage's decode step (w1|w3) IN SITU: rocprofv3 --kernel-trace --stats over a child
    that runs the stage's own step loop (insitu_stage_child); returns the roofline object of the N > 1 line, or (None, reason)."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    outdir = tempfile.mkdtemp(prefix="llamahip_stage_", dir="/tmp")
    steps = 48
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", outdir, "-o", "stage", "--",
           sys.executable, os.path.abspath(__file__), "--insitu-stage-child", "--model", model_name, "--n_ctx", str(args.n_ctx), "--stage-layers", str(stage_layers),
           "--stage-set", str(stage_set), "--steps", str(steps), "--warmup", "4", "--threads", str(args.threads), "--seed", str(args.seed)]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", LLAMAHIP_WITH_TORCH="1"), capture_output=True, text=True, timeout=300)
    except Exception as e:
        shutil.rmtree(outdir, ignore_errors=True)
        return None, repr(e)
    stats = glob.glob(os.path.join(outdir, "**", "*kernel_stats.csv"), recursive=True)
    if r.returncode != 0 or not stats:
        shutil.rmtree(outdir, ignore_errors=True)
        return None, f"rocprofv3 rc={r.returncode}: {r.stderr[-300:]}"
    rows = list(csv.DictReader(open(stats[0])))
    shutil.rmtree(outdir, ignore_errors=True)
    # w1|w3: the launch with the SiLU * up -> Q4_0 epilogue: for a set k_gemv_set<NC, CW, 7 | 2> (EPI_SILU_QAH, or EPI_SILU_QA from two column groups on),
    # k_gemv<.., 2 | 7, ..> for single steps
    pick = None
    for row in rows:
        n = row["Name"]
        n = (n[:n.index("(")] if "(" in n else n).replace("void ", "")
        last = n.rstrip(">").split(",")[-1].strip()
        hit = ("k_gemv_set<" in n and last in ("7", "2")) if stage_set > 1 else ("k_gemv<" in n and n.split("<")[1].split(",")[1].strip() in ("2", "7"))
        if hit and int(row["Calls"]) >= steps * stage_layers // 2:
            pick = (n, float(row["AverageNs"]) / 1e3, int(row["Calls"]))
    if not pick:
        return None, "no w1|w3 launch found in the stage's kernel statistics"
    d, F = cfg["n_embd"], n_ff(cfg)
    algo = 2 * F * (d // 32) * 20 + stage_set * (d // 32) * 20 + 4 * 2 * F * stage_set       # SURVEY.md 8d with N = stage_set activation rows
    gbps = algo / pick[1] / 1e3
    return {"bound": "hbm", "kernel": f"{pick[0]} -- the w1|w3 launch of a stage step as it runs in the stage's captured step ({stage_layers} layers, {stage_set} sequence(s) per step)",
            "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": algo,
            "us_per_launch": pick[1], "launches": pick[2],
            "method": "rocprofv3 --kernel-trace --stats of a child process running rank 0's stage loop (first-stage handle, same layer count and set size)"}, None


def parse_kernel_trace(outdir, cfg):
    """Label every dispatch of the decode loop by its position in the token's launch sequence (embedding -> per
    layer {wq|wk|wv, scores, soft_max * V, wo, w1|w3, w2} -> output -> argmax) and average the durations."""
    files = glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        return None
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            low = {k.lower(): v for k, v in r.items()}
            name = low.get("kernel_name", "")
            try:
                rows.append((int(low["start_timestamp"]), int(low["end_timestamp"]), name))
            except (KeyError, ValueError):
                continue
    rows.sort()
    layouts, out_role, nl = decode_roles(cfg)
    dur = {}
    names = {}
    ntok = 0
    i = 0
    while i < len(rows):
        matched = False
        if "k_embed" in rows[i][2]:
            for layer in layouts:
                seq_len = 1 + nl * len(layer) + 2
                if i + seq_len > len(rows) or "k_argmax" not in rows[i + seq_len - 1][2] or "k_gemv" not in rows[i + seq_len - 2][2]:
                    continue
                seg = rows[i:i + seq_len]
                if not all(sub in seg[1 + il * len(layer) + j][2] for il in range(nl) for j, (_, _, sub) in enumerate(layer)):
                    continue
                ntok += 1
                for il in range(nl):
                    for j, (role, _, _) in enumerate(layer):
                        s = seg[1 + il * len(layer) + j]
                        dur.setdefault(role, []).append((s[1] - s[0]) * 1e-3)
                        names[role] = s[2]
                for role, s in (("embed", seg[0]), (out_role[0], seg[seq_len - 2]), ("argmax", seg[seq_len - 1])):
                    dur.setdefault(role, []).append((s[1] - s[0]) * 1e-3)
                    names[role] = s[2]
                dur.setdefault("token_span", []).append((seg[-1][1] - seg[0][0]) * 1e-3)
                i += seq_len
                matched = True
                break
        if not matched:
            # the device-resident greedy loop since round 4: per layer launches, then the lm head's launch that also picks the token and
            # embeds it (EPI_STORE_PICK) -- no k_embed_part / k_argmax in a token.  (The window cannot lock onto a layer boundary inside a
            # token: the slot of its last layer's first launch would hold the lm head and the slot after it the NEXT token's first launch,
            # which is never the second launch of a layer.)
            for layer in layouts:
                seq_len = nl * len(layer) + 1
                if i + seq_len > len(rows) or "k_gemv" not in rows[i + seq_len - 1][2] or "k_embed" in rows[i][2]:
                    continue
                seg = rows[i:i + seq_len]
                if not all(sub in seg[il * len(layer) + j][2] for il in range(nl) for j, (_, _, sub) in enumerate(layer)):
                    continue
                ntok += 1
                for il in range(nl):
                    for j, (role, _, _) in enumerate(layer):
                        s_ = seg[il * len(layer) + j]
                        dur.setdefault(role, []).append((s_[1] - s_[0]) * 1e-3)
                        names[role] = s_[2]
                dur.setdefault(out_role[0], []).append((seg[-1][1] - seg[-1][0]) * 1e-3)
                names[out_role[0]] = seg[-1][2]
                dur.setdefault("token_span", []).append((seg[-1][1] - seg[0][0]) * 1e-3)
                i += seq_len
                matched = True
                break
        if not matched:
            i += 1
    if ntok == 0:
        return None
    short = lambda n: n[:n.index("(")] if "(" in n else n
    return {"tokens": ntok, "us": {k: float(np.mean(v)) for k, v in dur.items()}, "kernel": {k: short(v).replace("void ", "") for k, v in names.items()}}


def insitu_profile(args, cfg):
    """Run the decode loop under rocprofv3 --kernel-trace in a child process and return per-launch averages."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    outdir = tempfile.mkdtemp(prefix="llamahip_insitu_", dir="/tmp")
    steps = min(args.steps, 96)
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", outdir, "-o", "insitu", "--",
           sys.executable, os.path.abspath(__file__), "--insitu-child", "--model", args.model, "--n_ctx", str(args.n_ctx),
           "--steps", str(steps), "--warmup", str(args.warmup), "--threads", str(args.threads), "--seed", str(args.seed)]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    except Exception as e:          # the profile must never cost the headline line
        return None, repr(e)
    if r.returncode != 0:
        return None, f"rocprofv3 rc={r.returncode}: {r.stderr[-300:]}"
    prof = parse_kernel_trace(outdir, cfg)
    if args.save_profile and prof:
        stats = glob.glob(os.path.join(outdir, "**", "*kernel_stats.csv"), recursive=True)
        with open(args.save_profile, "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --insitu-child --model {args.model} --steps {steps} --warmup {args.warmup}"
                    f"   (the decode loop bench.py times, {prof['tokens']} tokens labelled by launch position)\n")
            f.write("# in-situ average per launch of the captured decode step, microseconds:\n")
            for k, v in prof["us"].items():
                f.write(f"#   {k:16s} {v:8.2f}   {prof['kernel'].get(k, '')}\n")
            if stats:
                rows = list(csv.DictReader(open(stats[0])))
                f.write(f"{'calls':>8} {'avg_us':>9} {'min_us':>8} {'max_us':>8} {'total_ms':>9} {'pct':>6}  kernel\n")
                for row in rows:
                    n = row["Name"]
                    n = n[:n.index("(")] if "(" in n else n
                    f.write(f"{int(row['Calls']):8d} {float(row['AverageNs']) / 1e3:9.2f} {float(row['MinNs']) / 1e3:8.2f} {float(row['MaxNs']) / 1e3:8.2f} "
                            f"{float(row['TotalDurationNs']) / 1e6:9.2f} {float(row['Percentage']):6.2f}  {n}\n")
    shutil.rmtree(outdir, ignore_errors=True)
    return prof, None if prof else "no decode token sequence found in the kernel trace"


def insitu_traffic(args, cfg):
    """HBM bytes per launch of the decode step's kernels, measured in THIS run: two rocprofv3 children over a short decode loop, one
    with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE (separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes).
    gfx950: FETCH_SIZE reports half the bytes of a wide streaming read -> traffic = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB.
    Returns ({kernel name prefix: bytes per launch}, None) or (None, reason)."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        outdir = tempfile.mkdtemp(prefix="llamahip_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--insitu-child", "--model", args.model, "--n_ctx", str(args.n_ctx),
               "--steps", "16", "--warmup", str(args.warmup), "--threads", str(args.threads), "--seed", str(args.seed)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
        except Exception as e:
            shutil.rmtree(outdir, ignore_errors=True)
            return None, repr(e)
        files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            shutil.rmtree(outdir, ignore_errors=True)
            return None, f"rocprofv3 --pmc {counter} rc={r.returncode}: {r.stderr[-200:]}"
        acc = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"]
                k = (k[:k.index("(")] if "(" in k else k).replace("void ", "")
                acc.setdefault(k, []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            if len(v) >= 16:                         # the decode loop's launches (a layer's kernels run >= 16 x n_layer times)
                res.setdefault(k, {})[counter] = sum(v) / len(v) * 1024.0
        shutil.rmtree(outdir, ignore_errors=True)
    out = {k: 2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0) for k, v in res.items() if "FETCH_SIZE" in v}
    return (out, None) if out else (None, "no decode launches in the counter files")


# ------------------------------------------------------------------------------------------------ the single-GPU run
def run_single(args, cfg, path):
    import llama_swift_amd as L
    import torch
    t0 = time.time()
    m = L.Model(path, n_ctx=args.n_ctx)
    t_load = time.time() - t0
    log(f"[bench] loaded in {t_load:.1f}s: {m.stats()}")
    prompt = PROMPT % cfg["n_vocab"]
    prompt[0] = 1
    logits = m.eval(prompt, 0, args.threads)
    first, n_past = int(np.argmax(logits)), len(prompt)
    steps = min(args.steps, args.n_ctx - n_past - args.warmup)
    tok, warm = first, np.zeros(0, np.int32)
    if args.warmup > 0:
        warm = m.decode_greedy(tok, n_past, args.warmup, args.threads)
        tok, n_past = int(warm[-1]), n_past + args.warmup
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.decode_greedy(tok, n_past, steps, args.threads)       # synchronises before returning
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_trace = [int(x) for x in warm] + [int(x) for x in out]     # every token generated after the prompt's pick `first` (extended by the full-context run below)
    # PCIe-inclusive variant: the reference boundary (llama_eval returning host logits every token); its tokens must
    # be those of the device-resident loop
    n_pcie = min(64, steps)
    t1 = time.perf_counter()
    tk, pcie_toks = tok, []
    for i in range(n_pcie):
        lg = m.eval(np.array([tk], np.int32), n_past + i, args.threads)
        tk = int(np.argmax(lg)); pcie_toks.append(tk)
    dt_pcie = time.perf_counter() - t1
    pcie_same = pcie_toks == [int(x) for x in out[:n_pcie]]
    # configs[1] is a 512-token generation: whatever --steps the caller passed, time the full context once more (8 prompt + 8 warm-up
    # + 496 generated tokens fill n_ctx 512); its tokens must continue the timed run's
    full = None
    n_full = args.n_ctx - len(prompt) - args.warmup
    if n_full > steps:
        t3 = time.perf_counter()
        out_full = m.decode_greedy(tok, len(prompt) + args.warmup, n_full, args.threads)
        dt_full = time.perf_counter() - t3
        full = {"steps": n_full, "seconds": dt_full, "tokens": out_full, "same_prefix": [int(x) for x in out_full[:steps]] == [int(x) for x in out]}
        if full["same_prefix"]:
            gpu_trace = [int(x) for x in warm] + [int(x) for x in out_full]      # the parity gate can follow the CPU path past a short --steps (configs[0]: 128 tokens)
    # stand-alone probe of the mat-vec kernel (PRE_QA / STORE variant, back-to-back launches cycling over the layers):
    # kept as a secondary figure -- this variant never runs in the decode step
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
    # ... and the same chunks in ONE pass that leaves the KV cache / logits of the chunk-by-chunk loop bit for bit (llamahip_eval_chunks:
    # what LlamaRunner.run does with a prompt)
    lg_loop = m.eval(ptoks[n9 - 9:n9], n9 - 9, args.threads) if n9 >= 9 else None
    lg_pass = m.eval_chunks(ptoks[:n9], 0, 9, args.threads)
    t2 = time.perf_counter(); m.eval_chunks(ptoks[:n9], 0, 9, args.threads); dt_9p = time.perf_counter() - t2
    prefill = {"one_eval": {"tokens": int(len(ptoks)), "tokens_per_s": len(ptoks) / dt_pre},
               "reference_9_token_chunks": {"tokens": int(n9), "tokens_per_s": n9 / dt_9},
               "reference_9_token_chunks_in_one_pass": {"tokens": int(n9), "tokens_per_s": n9 / dt_9p,
                                                        "logits_equal_chunk_by_chunk": bool(lg_loop is not None and np.array_equal(lg_loop, lg_pass))},
               "note": "exact path (bit-identical to the reference); host logits copy included"}

    def gpu_logits_at(n_tokens):
        """logits of the step that consumed generated token n_tokens - 1 (for the parity gate)."""
        lg0 = m.eval(prompt, 0, args.threads)
        t_ = int(np.argmax(lg0))
        _, last = m.decode_greedy(t_, len(prompt), n_tokens, args.threads, want_logits=True)
        return last
    res = dict(steps=steps, dt=dt, t_load=t_load, value_pcie=n_pcie / dt_pcie, pcie_same=pcie_same, shapes=shapes,
               gemv_bytes_per_token=tot_bytes, gemv_us_per_token=tot_us, prefill=prefill, first=first, gpu_trace=gpu_trace,
               n_past0=n_past, gpu_logits_at=gpu_logits_at, model=m, full=full)
    return res


def inprocess_pipeline(args, cfg, path, n_stages, single_trace):
    """single-stream greedy decode through ONE handle loaded with a device list (llamahip_opts.n_devices): n_stages stage handles in this
    process, all on device 0.  Tokens must be the single-device run's."""
    import llama_swift_amd as L
    with L.Model(path, n_ctx=args.n_ctx, devices=[0] * n_stages) as m:
        prompt = PROMPT % cfg["n_vocab"]
        prompt[0] = 1
        first = int(np.argmax(m.eval(prompt, 0, args.threads)))
        n = min(128, args.n_ctx - len(prompt) - 8)
        warm = m.decode_greedy(first, len(prompt), 8, args.threads)
        t0 = time.perf_counter()
        out = m.decode_greedy(int(warm[-1]), len(prompt) + 8, n, args.threads)
        dt = time.perf_counter() - t0
        got = [int(x) for x in warm] + [int(x) for x in out]
    return {"stages": n_stages, "devices": [0] * n_stages, "tokens": n, "tokens_per_s": n / dt, "ms_per_step": dt * 1e3 / n,
            "tokens_equal_single_device": got == [int(x) for x in single_trace[:len(got)]],
            "note": "one process, one llamahip_model_load; stage steps as captured graphs on one stream per stage, residual row and picked token handed on by stream-ordered copies"}


def inprocess_pipeline_multi(args, cfg, path, n_stages, n_seq, single_trace):
    """llamahip_decode_greedy_multi on a pipeline handle with every stage on device 0: n_seq sequences in groups (sets), the groups pipelined over
    the stages -- the native form of the multi-process schedule; sequence 0 decodes the bench prompt and must reproduce the single-stream tokens."""
    import llama_swift_amd as L
    with L.Model(path, n_ctx=args.n_ctx, devices=[0] * n_stages if n_stages > 1 else None, n_seq=n_seq) as m:
        rng = np.random.default_rng(99)
        prompts = [PROMPT % cfg["n_vocab"]] + [np.concatenate([[1], rng.integers(3, cfg["n_vocab"], 7 + s % 3)]).astype(np.int32) for s in range(1, n_seq)]
        prompts[0][0] = 1
        firsts = []
        for i, p in enumerate(prompts):
            m.set_seq(i)
            firsts.append(int(np.argmax(m.eval(p, 0, args.threads))))
        n_past = [len(p) for p in prompts]
        n = min(128, args.n_ctx - max(n_past) - 8)
        warm = m.decode_greedy_multi(firsts, n_past, 8, args.threads)
        t0 = time.perf_counter()
        out = m.decode_greedy_multi(warm[:, -1], [x + 8 for x in n_past], n, args.threads)
        dt = time.perf_counter() - t0
        got0 = [int(x) for x in warm[0]] + [int(x) for x in out[0]]
    return {"stages": n_stages, "sequences": n_seq, "steps": n, "aggregate_tokens_per_s": n_seq * n / dt, "ms_per_step": dt * 1e3 / n,
            "sequence0_tokens_equal_single_stream": got0 == [int(x) for x in single_trace[:len(got0)]]}


def concurrent_sequences(args, cfg, path, n_seq):
    """NOT the headline workload: `n_seq` independent greedy sequences decoded at the same time on ONE GPU -- one handle (its own weight
    copy, KV cache, stream and captured graph) and one host thread per sequence, no batching across sequences.  Every sequence is the
    headline's single-token decode; what changes is that one sequence's latency-bound phases (attention chain, norm prologues, launch
    ramps) are filled by the other sequences' weight streams.  Reports the aggregate rate and checks that every sequence produced the
    single-stream tokens."""
    import threading
    import llama_swift_amd as L
    prompt = PROMPT % cfg["n_vocab"]
    prompt[0] = 1
    steps = args.n_ctx - len(prompt) - 8
    ms = [L.Model(path, n_ctx=args.n_ctx) for _ in range(n_seq)]
    firsts, outs = [], [None] * n_seq
    for m in ms:
        firsts.append(int(np.argmax(m.eval(prompt, 0, args.threads))))
        m.decode_greedy(firsts[-1], len(prompt), 8, args.threads)            # captures the graph
    ref = ms[0].decode_greedy(firsts[0], len(prompt), steps, args.threads)   # single stream, same handle: the tokens to reproduce
    go = threading.Barrier(n_seq + 1)

    def work(i):
        go.wait()
        outs[i] = ms[i].decode_greedy(firsts[i], len(prompt), steps, args.threads)
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_seq)]
    for t in th:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    same = all(o is not None and [int(x) for x in o] == [int(x) for x in ref] for o in outs)
    for m in ms:
        m.close()
    return {"sequences": n_seq, "tokens": n_seq * steps, "seconds": dt, "aggregate_tokens_per_s": n_seq * steps / dt,
            "per_sequence_tokens_per_s": steps / dt, "tokens_equal_single_stream": same,
            "note": "independent sequences, one handle + host thread each, no cross-sequence batching; not the headline metric (one sequence)"}


def batched_sequences(args, cfg, path, n_seq):
    """NOT the headline workload: `n_seq` greedy sequences at different positions decoded by ONE handle with llamahip_stage_step_set --
    one decode step evaluates the next token of every sequence, so each weight matrix is streamed once per step for all of them
    (the pipeline's aggregate rate).  Bit-exact per sequence: every sequence must reproduce the tokens of its own
    single-stream decode."""
    import torch
    import llama_swift_amd as L
    m = L.Model(path, n_ctx=args.n_ctx, n_seq=n_seq)
    prompts, firsts, refs = [], [], []
    steps = args.n_ctx - (len(PROMPT) + n_seq) - 8
    for s in range(n_seq):
        pr = np.roll(PROMPT % cfg["n_vocab"], s)[: len(PROMPT)]
        pr = np.concatenate([pr, (np.arange(s, dtype=np.int64) * 977 + 5) % cfg["n_vocab"]]).astype(np.int32)      # lengths 8, 9, ...: every sequence at its own position
        pr[0] = 1
        prompts.append(pr)
        m.set_seq(s)
        firsts.append(int(np.argmax(m.eval(pr, 0, args.threads))))
        refs.append([int(x) for x in m.decode_greedy(firsts[-1], len(pr), steps, args.threads)])      # single stream, same handle and cache: the tokens to reproduce
    m.set_seq(0)
    bufs = [torch.tensor([firsts[s]], dtype=torch.int32, device="cuda") for s in range(n_seq)]
    st = torch.cuda.current_stream().cuda_stream
    seqs = list(range(n_seq))

    def bind():
        for s in range(n_seq):
            bufs[s].fill_(firsts[s])
            m.stage_bind(s, len(prompts[s]), token_in=bufs[s].data_ptr(), token_out=bufs[s].data_ptr())
    bind()
    for _ in range(8):
        m.stage_step_set(seqs, args.threads, st)          # captures the graph
    torch.cuda.synchronize()
    bind()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.stage_step_set(seqs, args.threads, st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = True
    for s in range(n_seq):
        n, pos, got = m.stage_trace(s, steps)
        same = same and n == steps and [int(x) for x in got] == refs[s]
    m.close()
    wb = token_bytes(cfg, 0)
    return {"sequences": n_seq, "tokens": n_seq * steps, "seconds": dt, "aggregate_tokens_per_s": n_seq * steps / dt, "ms_per_step": dt * 1e3 / steps,
            "per_sequence_tokens_per_s": steps / dt, "tokens_equal_single_stream": same,
            "weights_once_per_step_frac": wb / (dt / steps) / 1e9 / HBM_PEAK_GBPS,
            "note": "one llamahip_stage_step_set per step: the few-row kernels (k_gemv_set, per-row attention) with per-row position / KV cache / V*P key split; "
                    "not the headline metric (one sequence)"}


# ------------------------------------------------------------------------------------------------------------------------------
# N > 1: the layer pipeline (llama_swift_amd/pipeline.py holds the schedules; the bench leg and its CPU parity checker live here)
# ------------------------------------------------------------------------------------------------------------------------------
def _cpu_trace(path: str, prompt: np.ndarray, n_tokens: int, n_ctx: int, budget_s: float):
    """Greedy tokens of the CPU path (the reference's ggml.c build when it travelled with the snapshot, else the restatement) for one
    prompt: the parity gate of the multi-GPU line.  Bounded by `budget_s` seconds of decoding.  Checker only, after the timed loops."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import reflib
    kind = "reference" if reflib.have_ref() else "port"
    lib = reflib.RefLib() if kind == "reference" else reflib.OracleLib()
    m = lib.load(path, n_ctx, 0)
    m.eval(np.array([0, 1, 2, 3], np.int32), 0, 8)            # the bridge's scratch-sizing eval (.mm:820-822)
    lg = m.eval(prompt, 0, 8)["logits"]
    t, toks, n_past = int(np.argmax(lg)), [], len(prompt)
    first = t
    t0 = time.time()
    while len(toks) < n_tokens and time.time() - t0 < budget_s:
        lg = m.eval(np.array([t], np.int32), n_past, 8)["logits"]
        t = int(np.argmax(lg)); toks.append(t); n_past += 1
    m.close()
    return kind, first, toks


class CudaEnv:
    """Where the pipeline bench runs: one MI355X per rank, RCCL.  (tests/test_pipeline.py drives the same control flow on the CPU with
    gloo and oracle stages through an environment of the same shape.)"""
    backend_default = "nccl"

    def __init__(self, local):
        import torch
        self.torch, self.local, self.device = torch, local, f"cuda:{local}"
        torch.cuda.set_device(local)

    def init_process_group(self, dist, backend):
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=self.torch.device(self.device))
        else:
            dist.init_process_group(backend)

    def sync(self):
        self.torch.cuda.synchronize()

    def lane(self):
        return self.torch.cuda.Stream()

    def on(self, lane):
        return self.torch.cuda.stream(lane)

    def make_stage(self, path, n_ctx, rank, world, n_seq, n_layer, n_threads):
        from llama_swift_amd.pipeline import HipStage
        return HipStage(path, n_ctx, rank, world, n_seq, n_layer, self.local, n_threads)

    def stage_roofline(self, stage, args, cfg, model_name=None, stage_layers=0, stage_set=1, rank=0):
        """The dominant launch of rank 0's stage IN SITU: a rocprofv3 child runs the stage's own step loop (same layer count, same set
        size: the stage's launches, kernels and shapes); falls back to the stand-alone probe and says so."""
        if rank != 0:
            return None
        if model_name and stage_layers > 0 and os.environ.get("LLAMAHIP_PIPE_NO_INSITU") != "1":
            roof, why = insitu_stage_profile(args, model_name, cfg, stage_layers, stage_set)
            if roof:
                roof["per_stage_weight_bytes"] = stage.model.stats()["weight_bytes_device"]
                return roof
            log(f"[bench] in-situ stage profile unavailable ({why}); stand-alone probe")
        try:
            r = stage.model.bench_gemv(2, -1, 1, 10)
            return {"bound": "hbm", "kernel": "lh::k_gemv PRE_QA / STORE probe variant on w1|w3 of rank 0's layers (stand-alone, not in situ)", "achieved": r["GBps"], "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": r["GBps"] / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": r["algo_bytes"], "us_per_launch": r["us_per_launch"],
                    "per_stage_weight_bytes": stage.model.stats()["weight_bytes_device"]}
        except Exception as e:                       # measurement extras never cost the headline line
            return {"error": repr(e)}

    def close_stage(self, stage):
        stage.model.close()


def pipeline_bench_main(args, cfg, model_path_fn, log, models=None, env=None, emit=None):
    """`bench.py --gpus N` for N > 1 (launched by torch.distributed.run, one rank per GPU).  `env` / `emit`: the CPU control-flow test
    substitutes gloo + oracle stages and collects the line instead of printing it."""
    import threading

    import torch
    import torch.distributed as dist

    from llama_swift_amd.pipeline import ERR_PREDICT, gather_traces, layer_range, mailbox_decode, make_groups, pipeline_decode, pipeline_decode_sets, pipeline_rounds, run_guarded, transport_selfcheck
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} must be launched with {args.gpus} ranks (WORLD_SIZE={world}); "
                         f"use: python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}")
    in_test = env is not None
    # RCCL prints a version banner on STDOUT when it creates a communicator; the bench contract is ONE
    # JSON line on stdout, so everything before that line goes to stderr at the file-descriptor level
    saved_stdout = None
    if not in_test:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)

    # a multi-rank run that stops making progress (a peer died, a hand-off never matched) must not hang the
    # caller forever: every schedule call below runs under run_guarded (transport error -> PipelineError, silence ->
    # exit code 3 after `limit` seconds), and the whole bench under one more timer of the same length
    limit = float(os.environ.get("LLAMAHIP_PIPE_WATCHDOG_S", "900"))
    headline = {}                                # rank 0: the finished JSON line of the headline model (printed by whoever ends the run)
    headline_done = [False]                      # EVERY rank: the headline leg is over (what follows is the optional 65B leg)

    def _emit_and_exit(code):
        if rank == 0 and headline and saved_stdout is not None:
            os.dup2(saved_stdout, 1)
            os.write(1, (json.dumps(headline) + "\n").encode())
        # a stuck EXTRA leg must not cost the headline that is already measured -- on ANY rank: a non-zero exit of rank r > 0 makes the
        # launcher kill rank 0, possibly before it has printed
        os._exit(0 if headline_done[0] else code)

    def _abort():
        os.write(2, f"[bench] rank {rank}/{world}: no result after {limit:.0f} s -- PredictionFailed ({ERR_PREDICT}), aborting\n".encode())
        _emit_and_exit(3)

    watchdog = threading.Timer(limit, _abort)
    watchdog.daemon = True
    watchdog.start()
    guard = lambda fn, what: run_guarded(fn, rank, world, limit, what, on_timeout=(lambda msg: _emit_and_exit(3)) if headline_done[0] else None)
    # (smoke test of the multi-rank path on ONE GPU: LLAMAHIP_PIPE_ONE_GPU=1 puts every rank on cuda:0 and LLAMAHIP_PIPE_BACKEND=gloo
    #  replaces RCCL, which refuses two ranks on one device; the mailboxes then run over HIP IPC between the processes)
    if os.environ.get("LLAMAHIP_PIPE_ONE_GPU") == "1":
        local = 0
    hsync = None
    if env is None:
        for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):      # (LLAMAHIP_FORCE_PIPELINE: one rank without a launcher)
            os.environ.setdefault(k_, v_)
        env = CudaEnv(local)
        backend = os.environ.get("LLAMAHIP_PIPE_BACKEND", env.backend_default)
        env.init_process_group(dist, backend)
        if backend != "nccl":
            hsync = env.sync                     # gloo moves CUDA tensors through the host, not in device-stream order: synchronise around it (smoke runs only)
    dev = env.device
    backend_ran = dist.get_backend()
    transport = "RCCL" if backend_ran == "nccl" else f"{backend_ran} (smoke transport, host-synchronised around every message)"
    # the token feedback edge on its own communicator, the forward edges on two by sender parity (pipeline.make_groups: the same function
    # under gloo in the CPU tests), then one barrier / object all-gather / all-reduce per communicator before anything is timed
    token_group, fwd_groups, topology = make_groups(dist, world)
    selfcheck = transport_selfcheck(dist, torch, rank, world, dev, backend_ran, token_group, fwd_groups, log)
    sync_schedule = os.environ.get("LLAMAHIP_PIPELINE_SYNC", "0") == "1"
    # sequences per stage: a stage steps them as ONE set (llamahip_stage_step_set: its weights are streamed once per step for all of
    # them) and the set's rows cross to the next stage in one message.  LLAMAHIP_PIPE_SET=0: one sequence per step, device-side
    # mailboxes between the stages (round 3's schedule; the single-stream latency leg always runs one sequence per step).
    per_stage = max(1, int(os.environ.get("LLAMAHIP_PIPE_SEQS_PER_STAGE", "8")))      # (8: 2 930 against 1 965 tokens/s for sets of 4 on one GPU, profiles/r05_*)
    set_mode = os.environ.get("LLAMAHIP_PIPE_SET", "1") != "0" and per_stage >= 2 and not sync_schedule
    want_mailbox = os.environ.get("LLAMAHIP_PIPE_MAILBOX", "1") != "0" and not sync_schedule and world > 1 and not set_mode

    def run_model(model_name, mcfg, steps_req, warmup, parity_tokens):
        if rank == 0:
            model_path_fn(model_name, mcfg, args.seed)
        dist.barrier()
        path = model_path_fn(model_name, mcfg, args.seed)
        # sequences in flight (weak scaling): `per_stage` per stage.  One group of `per_stage` consecutive slots per stage in set mode;
        # with one sequence per step a second one per stage keeps a ready item queued behind the hand-off latency.
        S = world * per_stage if (world > 1 or set_mode) else 1
        if not set_mode and world > 1:
            S = world * min(per_stage, 2) if "LLAMAHIP_PIPE_SEQS_PER_STAGE" not in os.environ else world * per_stage
        groups = [list(range(g * per_stage, (g + 1) * per_stage)) for g in range(S // per_stage)] if set_mode else None
        stage = env.make_stage(path, args.n_ctx, rank, world, S, mcfg["n_layer"], args.threads)
        rng = np.random.default_rng(1234)
        prompts = [np.concatenate([[1], rng.integers(3, mcfg["n_vocab"], 7 + (s % 3 if set_mode else 0))]).astype(np.int32) for s in range(S)]   # (set mode: rows of a set at different positions)
        n_single = 16                                          # single-stream latency leg: tokens of sequence 0 alone
        steps = max(1, min(steps_req, args.n_ctx - 11 - warmup - 1 - n_single))
        hand_off = f"{transport} point-to-point per token (torch.distributed isend / recv, stream-ordered)"
        if set_mode:
            hand_off = f"{transport} point-to-point, one message per set of {per_stage} sequences and step (torch.distributed isend / recv, stream-ordered)"
        if sync_schedule:
            toks, n_past = guard(lambda: pipeline_rounds(stage, rank, world, dist, torch, prompts, [0] * S, 1 + warmup, token_group), "pipeline_rounds (prompt + warm-up)")
            last = [np.array([toks[s, -1]], np.int32) for s in range(S)]
            dist.barrier(); env.sync()
            t0 = time.perf_counter()
            toks2, n_past = guard(lambda: pipeline_rounds(stage, rank, world, dist, torch, last, n_past, steps, token_group), "pipeline_rounds (timed decode)")
            dist.barrier(); env.sync()
            dt_loc = time.perf_counter() - t0
            firsts = [int(toks[s, 0]) for s in range(S)]
            traces = np.concatenate([toks[:, 1:], toks2], axis=1)
            single = None
        else:
            toks, n_past = guard(lambda: pipeline_rounds(stage, rank, world, dist, torch, prompts, [0] * S, 1, token_group), "pipeline_rounds (prompt)")
            firsts = [int(toks[s, -1]) for s in range(S)]
            mailbox = False
            if want_mailbox:
                # device-side mailboxes: one object all-gather of IPC handles now, no collective per token afterwards.  Every rank
                # must take the same branch: agree on the outcome.
                ok = 1
                try:
                    stage.setup_mailboxes(dist, rank, world, S)
                except Exception as e:                          # e.g. IPC not permitted on this box
                    log(f"[bench] rank {rank}: mailboxes unavailable ({type(e).__name__}: {e}); RCCL hand-off")
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                mailbox = int(flag.item()) == 1
                if not mailbox:
                    stage.mailboxes = False
            for s in range(S):
                stage.bind(s, n_past[s], firsts[s])
            # binding clears a slot's inbox: no stage may store into its neighbour's before every stage has bound
            env.sync(); dist.barrier()
            lane = env.lane()                        # the decode loop's own stream
            if mailbox:
                # handshake: ONE token of sequence 0 through every stage on the mailboxes, then every rank reads its fault word.  A
                # row that does not arrive (peer mapping that does not carry stores, ...) costs one poll bound here, not one per step
                # of the timed loop; all ranks agree on the outcome and fall back to the RCCL hand-off together.
                ok = 1
                try:
                    with env.on(lane):
                        stage.step(0)
                    env.sync()
                    n_done, pos0, _ = stage.trace(0, 1)
                    ok = int(n_done == 1 and pos0 == n_past[0] + 1)
                except Exception as e:
                    log(f"[bench] rank {rank}: mailbox handshake failed ({type(e).__name__}: {str(e)[:200]}); RCCL hand-off")
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) != 1:
                    mailbox = False
                    stage.mailboxes = False
                    toks, n_past = guard(lambda: pipeline_rounds(stage, rank, world, dist, torch, prompts, [0] * S, 1, token_group), "pipeline_rounds (prompt, again)")
                    firsts = [int(toks[s, -1]) for s in range(S)]
                    for s in range(S):
                        stage.bind(s, n_past[s], firsts[s])
                    env.sync(); dist.barrier()
            handshake_tokens = 1 if mailbox else 0
            if mailbox:
                hand_off = "device-side mailboxes: position-tagged granules stored into the next stage's memory (HIP IPC / xGMI) by the last kernel of a stage step, polled by the first kernel of the next; no collective and no host call per token"

            def decode(n, seqs=None):
                with env.on(lane):
                    if mailbox:
                        mailbox_decode(stage, S, n, seqs)
                    elif set_mode and seqs is None:
                        pipeline_decode_sets(stage, rank, world, dist, groups, n, fwd_groups, token_group, hsync)
                    else:
                        pipeline_decode(stage, rank, world, dist, S if seqs is None else len(seqs), n, fwd_groups, token_group, hsync)
                env.sync()
            guard(lambda: decode(warmup), "decode (warm-up)")                                # untimed; captures the graphs
            dist.barrier(); env.sync()
            t0 = time.perf_counter()
            guard(lambda: decode(steps), "decode (timed)")
            dist.barrier(); env.sync()
            dt_loc = time.perf_counter() - t0
            # single-stream latency, measured: sequence 0 alone through all stages
            single = None
            if world > 1:
                dist.barrier(); env.sync()
                t1 = time.perf_counter()
                guard(lambda: decode(n_single, [0]), "decode (single stream)")
                dist.barrier(); env.sync()
                ds = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
                dist.all_reduce(ds, op=dist.ReduceOp.MAX)
                single = {"tokens": n_single, "ms_per_token": float(ds.item()) * 1e3 / n_single, "tokens_per_s": n_single / float(ds.item()),
                          "note": "sequence 0 alone: one token at a time through every stage (the latency a single user sees)"}
            traces, _pos = guard(lambda: gather_traces(stage, rank, world, dist, torch, S, handshake_tokens + warmup + steps + (n_single if world > 1 else 0)), "gather_traces")
        dt = torch.tensor([dt_loc], dtype=torch.float64, device=dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dt = float(dt.item())
        # parity gate: sequence 0's prompt pick and first generated tokens against the CPU path (rank 0 computes it, bounded);
        # in set mode also the last sequence of the last set (another row of a batched step, another position)
        parity = {"checked": False}
        if rank == 0 and parity_tokens > 0:
            try:
                budget = float(os.environ.get("LLAMAHIP_PIPE_PARITY_S", "30"))
                checked = []
                for sq in ([0, S - 1] if (set_mode and S > 1) else [0]):
                    kind, cfirst, ctoks = _cpu_trace(path, prompts[sq], parity_tokens, args.n_ctx, budget / (2 if set_mode and S > 1 else 1))
                    have = (0 if sync_schedule else handshake_tokens) + warmup + steps + (n_single if (world > 1 and sq == 0 and not sync_schedule) else 0)      # tokens this sequence generated
                    ctoks = ctoks[:have]
                    got = [int(t) for t in traces[sq][:len(ctoks)]]
                    checked.append({"sequence": sq, "prompt_pick_identical": cfirst == firsts[sq], "tokens_compared": len(ctoks), "identical": cfirst == firsts[sq] and got == ctoks,
                                    "first_divergence": next((i for i, (x, y) in enumerate(zip(got, ctoks)) if x != y), None)})
                parity = {"checked": True, "against": f"{kind} CPU path, 8 threads, sequence(s) {[c['sequence'] for c in checked]}",
                          "prompt_pick_identical": all(c["prompt_pick_identical"] for c in checked), "tokens_compared": sum(c["tokens_compared"] for c in checked),
                          "identical": all(c["identical"] for c in checked), "first_divergence": next((c["first_divergence"] for c in checked if c["first_divergence"] is not None), None),
                          "per_sequence": checked}
            except Exception as e:                              # the checker must never take the measurement down
                parity = {"checked": False, "error": repr(e)}
        lo_, hi_ = layer_range(mcfg["n_layer"], 0, world)
        roof = env.stage_roofline(stage, args, mcfg, model_name, hi_ - lo_, per_stage if set_mode else 1, rank)
        env.close_stage(stage)
        return dict(S=S, steps=steps, dt=dt, parity=parity, roof=roof, single=single, hand_off=hand_off, n_layer=mcfg["n_layer"], set_mode=set_mode)

    r = run_model(args.model, cfg, args.steps, args.warmup, 8)
    headline_done[0] = True
    if rank == 0:
        total = r["S"] * r["steps"]
        headline.update({
            "metric": f"decode tokens/sec LLaMA-{args.model} Q4_0 @{world} GPUs (layer pipeline, {r['S']} sequences in flight); % HBM-roofline on Q4_0 GEMV",
            "value": total / r["dt"], "unit": "tokens/s", "n_gpus": world, "steps": r["steps"], "warmup": args.warmup,
            "ms_per_step": r["dt"] * 1e3 / r["steps"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "q4_0 x q4_0 -> int32 block sums, fp32 scales/accumulate",
            "data": "synthetic (random-init weights in the reference file format, synthetic token ids)",
            "config": {"workload": f"LLaMA-{args.model} Q4_0 greedy decode, {world}-stage layer pipeline "
                                   f"({r['n_layer']} layers / {world}), {r['S']} independent sequences in flight"
                                   + (f" ({per_stage} per stage, stepped as one set: llamahip_stage_step_set)" if r["set_mode"] else "") + f", n_ctx {args.n_ctx}; "
                                   f"a step = one token for every sequence",
                       "parallelism": f"pp{world}", "hand_off": r["hand_off"], "backend": backend_ran,
                       "transport_selfcheck": f"barrier + all_gather_object + all_reduce on 3 communicators ok on {len(selfcheck)} rank(s)",
                       "group_topology": topology,
                       "sequences": r["S"], "tokens_timed": total},
            "roofline": r["roof"],
            "parity": r["parity"],
            "cpu_baseline": None,
            "cpu_baseline_note": "timed at N = 1 only (bench.py --gpus 1)",
            "single_stream": r["single"],
            "schedule": "host-synchronous" if sync_schedule else "stream-ordered (hipGraph stage steps, device-side greedy pick)",
        })
    # BASELINE.json configs[4]: the 65B model is what the 8-GPU pipeline is for.  A bounded extra leg (32 timed steps) with its OWN
    # timer, reported next to the headline; whatever happens to it -- on any rank -- the headline above is printed and every rank exits 0.
    if models and args.model != "65B" and os.environ.get("LLAMAHIP_BENCH_65B", "1") != "0" and "65B" in models:
        leg_limit = float(os.environ.get("LLAMAHIP_PIPE_65B_S", "420"))
        leg_timer = threading.Timer(leg_limit, lambda: (os.write(2, f"[bench] rank {rank}: the 65B leg did not finish in {leg_limit:.0f} s; the headline stands\n".encode()), _emit_and_exit(0)))
        leg_timer.daemon = True
        leg_timer.start()
        ok65, r65, e65 = 1, None, None
        try:
            r65 = run_model("65B", models["65B"], int(os.environ.get("LLAMAHIP_PIPE_65B_STEPS", "32")), 4, 4)
        except BaseException as e:                   # (SystemExit from a guard included: the headline survives)
            ok65, e65 = 0, e
            import traceback
            os.write(2, f"[bench] rank {rank}: the 65B leg failed: {e!r}\n{traceback.format_exc()[-1500:]}\n".encode())
        leg_timer.cancel()
        if rank == 0:
            if ok65:
                t65 = r65["S"] * r65["steps"]
                headline["config4_65B"] = {"workload": f"LLaMA-65B Q4_0, {r65['n_layer']} layers over {world} stages, {r65['S']} sequences in flight",
                                           "tokens_per_s": t65 / r65["dt"], "ms_per_step": r65["dt"] * 1e3 / r65["steps"], "steps": r65["steps"],
                                           "parity": r65["parity"], "single_stream": r65["single"], "hand_off": r65["hand_off"], "roofline": r65["roof"]}
            else:
                headline["config4_65B"] = {"error": repr(e65)}
        if not ok65:
            # this rank left the leg early: its peers may sit in a collective of it until their guards fire.  Do not enter another
            # collective (destroy_process_group) with them: print and leave.
            watchdog.cancel()
            if emit is not None and rank == 0:
                emit(headline)
            _emit_and_exit(0)
    watchdog.cancel()
    if emit is not None:
        if rank == 0:
            emit(headline)
        return
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(headline), flush=True)
    os.dup2(2, 1)                                # communicator teardown may print as well
    dist.destroy_process_group()


def prefill_2048(args, cfg, path):
    """configs[2]: one 2048-token eval at n_ctx 2560 (second handle), with the rates the north star asks for."""
    import llama_swift_amd as L
    n_ctx, N = 2560, 2048
    m = L.Model(path, n_ctx=n_ctx)
    rng = np.random.default_rng(9)
    ptoks = rng.integers(3, cfg["n_vocab"], N).astype(np.int32)
    ptoks[0] = 1
    m.eval(ptoks, 0, args.threads)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); lg = m.eval(ptoks, 0, args.threads); best = min(best, time.perf_counter() - t0)
    # the same prompt the way the reference's driver feeds it: nine tokens per llama_eval, here as one chunk-exact pass (llamahip_eval_chunks)
    flow = None
    try:
        m.eval_chunks(ptoks, 0, 9, args.threads)
        t0 = time.perf_counter(); lg = m.eval_chunks(ptoks, 0, 9, args.threads); dtf = time.perf_counter() - t0
        flow = {"tokens": N, "seconds": dtf, "tokens_per_s": N / dtf, "note": "KV cache and logits bit for bit those of 228 successive 9-token evals"}
    except Exception as e:
        flow = {"error": repr(e)}
    # ... and the decode that follows such a prompt: 64 greedy tokens from position 2048 (the long-context attention schedule,
    # DESIGN.md section 9.5; round 2: 480 tokens/s on the 7B)
    after = None
    try:
        tok, nd = int(np.argmax(lg)), 64
        m.decode_greedy(tok, N, 4, args.threads)
        bd = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); m.decode_greedy(tok, N, nd, args.threads); bd = min(bd, time.perf_counter() - t0)
        after = {"context": [N, N + nd], "tokens_per_s": nd / bd, "ms_per_token": bd / nd * 1e3,
                 "e2e_frac_of_hbm_peak": sum(token_bytes(cfg, N + i) for i in range(nd)) / bd / 1e9 / HBM_PEAK_GBPS}
    except Exception as e:                                  # never at the cost of the prefill numbers
        after = {"error": repr(e)}
    m.close()
    d, F, V, Lr = cfg["n_embd"], n_ff(cfg), cfg["n_vocab"], cfg["n_layer"]
    useful = 2.0 * N * Lr * (4 * d * d + 3 * d * F) + 2.0 * V * d            # mat-mul ops, last row of the lm head only
    fma_flops = useful / 4.0                                                  # the exact path: 8 fp32 FMAs per 32-element block and output
    return {"tokens": N, "n_ctx": n_ctx, "seconds": best, "tokens_per_s": N / best, "in_reference_9_token_chunks_one_pass": flow, "decode_after_prompt": after,
            "roofline": {"useful_ops": useful, "useful_TOPs": useful / best / 1e12,
                         "f16_mfma_peak_TFLOPs": F16_MFMA_PEAK_TFLOPS, "frac_of_f16_mfma_peak": useful / best / 1e12 / F16_MFMA_PEAK_TFLOPS,
                         "int8_mfma_peak_TOPs": INT8_MFMA_PEAK_TOPS, "frac_of_int8_mfma_peak": useful / best / 1e12 / INT8_MFMA_PEAK_TOPS,
                         "fp32_fma_TFLOPs": fma_flops / best / 1e12, "fp32_valu_peak_TFLOPs": FP32_VALU_PEAK_TFLOPS,
                         "frac_of_fp32_valu_peak": fma_flops / best / 1e12 / FP32_VALU_PEAK_TFLOPS,
                         "bound": "fp32 VALU issue: the reference's arithmetic needs 8 separately rounded fp32 FMA chains per Q4_0 block and "
                                  "output (ggml.c:1415-1466), 3.3e12 plain v_fma_f32 for this eval.  Only the 4-element integer sums of a chain "
                                  "run on the matrix cores (exact in fp16 -> fp32: v_mfma_f32_16x16x4_4b_f16, four chains per issue, "
                                  "lh::k_gemm_mfma4 at four waves per SIMD).  On gfx950 an MFMA and the VALU instructions of other waves of the same SIMD "
                                  "do not run side by side (tools/mfma_overlap_probe.hip, profiles/r04_p_mfma_overlap.txt): per Q4_0 block and 16 x 16 "
                                  "sub-tile the SIMD spends the MFMA's 14 ns PLUS 16 FMAs (19 ns) and their operand preparation -- 39 ns measured, so "
                                  "the fp32 FMA rate the path can reach is about half the VALU peak; the whole eval (attention, norms, quantizers "
                                  "included) is divided by the peaks here"}}


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
    ap.add_argument("--no-insitu", action="store_true", help="skip the rocprofv3 child that times the launches of the decode step")
    ap.add_argument("--no-prefill-2048", action="store_true")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the concurrent-sequences leg (2 and 4 independent sequences on the one GPU)")
    ap.add_argument("--save-profile", default="", help="write the in-situ kernel table (rocprofv3 summary) to this file")
    ap.add_argument("--insitu-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--insitu-stage-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stage-layers", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--stage-set", type=int, default=1, help=argparse.SUPPRESS)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cfg = MODELS[args.model]

    if args.insitu_child:
        return insitu_child(args, cfg, model_path(args.model, cfg, args.seed))
    if args.insitu_stage_child:
        return insitu_stage_child(args, cfg, model_path(args.model, cfg, args.seed))

    if world > 1 or args.gpus > 1 or os.environ.get("LLAMAHIP_FORCE_PIPELINE"):      # FORCE: exercise the N > 1 code path on one GPU
        # every stream of a rank (compute, one per RCCL communicator) gets its own hardware queue, so
        # a point-to-point kernel waiting for its peer can never sit in front of unrelated work
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        return pipeline_bench_main(args, cfg, model_path, log, MODELS)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the HIP path)")
    path = model_path(args.model, cfg, args.seed)
    r = run_single(args, cfg, path)
    m = r.pop("model")
    tps = r["steps"] / r["dt"]
    ms_step = r["dt"] * 1e3 / r["steps"]

    # ---- parity gate + CPU baselines (the CPU path decodes the same prompt: its tokens check the timed run)
    parity = {"checked": False}
    cpu_base = None
    if not args.no_cpu_baseline and args.cpu_seconds > 0:
        try:
            # the timed run's tokens, and configs[0]'s 128 greedy tokens whenever the budget allows (~7 s of the reference's 8-thread path;
            # the driver's --steps 20 would otherwise stop the CPU side at 25 tokens)
            want = args.warmup + r["steps"]
            if args.cpu_seconds >= 10:
                want = min(max(want, 128), len(r["gpu_trace"]))
            c = cpu_decode(path, 8, args.cpu_seconds, args.n_ctx, want)
            n = min(len(c["toks"]), len(r["gpu_trace"]))
            bad = [i for i in range(n) if c["toks"][i] != r["gpu_trace"][i]]
            dl = None
            if n > 0 and not bad and c["first"] == r["first"]:
                dl = float(np.abs(r["gpu_logits_at"](n) - c["last_logits"]).max())
            parity = {"checked": True, "against": c["kind"] + " CPU path, 8 threads", "tokens_compared": n, "tokens_in_timed_run_covered": max(0, n - args.warmup),
                      "timed_run_fully_covered": n >= args.warmup + r["steps"], "configs0_128_tokens_covered": n >= 128, "identical": not bad and c["first"] == r["first"], "first_divergence": bad[0] if bad else None,
                      "max_abs_dlogit_at_last_compared_step": dl, "tolerance": 1e-3,
                      "pcie_loop_tokens_equal_device_loop": r["pcie_same"]}
            cpu_base = {"value": len(c["toks"]) / c["dt"], "unit": "tokens/s", "cores": 8, "kind": c["kind"], "tokens": len(c["toks"]),
                        "covers_configs0_128_tokens": len(c["toks"]) >= 128,
                        "sample": f"{len(c['toks'])} greedy decode tokens after an 8-token prompt, same synthetic model file, {c['dt']:.1f}s wall, "
                                  f"8 threads (reference default numThreads=8, Sources/llama/LlamaRunner.swift:17)",
                        "host_cpus": os.cpu_count()}
            nall = min(os.cpu_count() or 8, 64)
            if nall > 8:
                c2 = cpu_decode(path, nall, min(8.0, args.cpu_seconds), args.n_ctx, 10 ** 9)
                cpu_base["all_cores"] = {"value": len(c2["toks"]) / c2["dt"], "unit": "tokens/s", "cores": nall,
                                         "sample": f"{len(c2['toks'])} tokens, {c2['dt']:.1f}s wall, n_threads = min(nproc, 64) = {nall} "
                                                   f"(the thread pool busy-spins, ggml.c:9061-9107; more threads than this only slow it down)"}
        except Exception as e:  # the checker must never take the measurement down
            cpu_base = {"value": None, "error": repr(e)}
    m.close()

    # ---- roofline of the decode step's launches, in situ
    layouts, out_role, nl = decode_roles(cfg)
    role_bytes = dict([(k, v) for layer_roles in layouts for k, v, _ in layer_roles if v] + [out_role])      # (only the roles of the layout that ran appear in the profile)
    prof, prof_err = (None, "skipped") if args.no_insitu else insitu_profile(args, cfg)
    probe = {"note": "stand-alone k_gemv<PRE_QA, EPI_STORE> launched back to back over all layers (HIP events); never runs in the decode step",
             "per_shape": [{k: s[k] for k in ("name", "M", "K", "us_per_launch", "GBps")} for s in r["shapes"]],
             "all_gemv_launches_of_a_token": {"GBps": r["gemv_bytes_per_token"] / (r["gemv_us_per_token"] * 1e-6) / 1e9}}
    t_mid = r["n_past0"] + r["steps"] // 2
    e2e_bytes = token_bytes(cfg, t_mid)
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "end_to_end": {"bytes_per_token": e2e_bytes, "at_position": t_mid, "GBps": e2e_bytes / (ms_step * 1e-3) / 1e9,
                           "frac": e2e_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                           "note": "W + KV(t) + small of SURVEY.md 8d at the middle position of the timed run / ms_per_step / 8 TB/s"},
            "end_to_end_frac": e2e_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "probe_back_to_back": probe}
    traffic = None
    traffic_src = None
    pmc, pmc_err = (None, "skipped") if args.no_insitu else insitu_traffic(args, cfg)
    try:        # fallback only: figures committed under profiles/ (labelled as such in the output)
        tj = json.load(open(os.path.join(ROOT, "profiles", "gemv_traffic.json")))
        traffic = tj["per_launch"]["w1|w3"]["traffic_bytes"] if args.model == "7B" else None
        traffic_src = "profiles/gemv_traffic.json (separate rocprofv3 --pmc passes over the stand-alone probe launches, committed; not measured in this run)"
        try:        # preferred: the same counters over the captured decode step itself (tools/pmc_decode_pass.sh)
            dj = json.load(open(os.path.join(ROOT, "profiles", "decode_traffic.json")))
            hit = [v for k, v in dj["per_launch"].items() if k.startswith("lh::k_gemv<4, 2, 4")]
            if hit and args.model == "7B":
                traffic = hit[0]["traffic_bytes"]
                traffic_src = "profiles/decode_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the captured decode step, in situ; committed, not measured in this run)"
        except Exception:
            pass
    except Exception:
        pass
    traffic_committed = traffic
    traffic = None
    if prof:
        per = []
        gb = gu = 0.0
        for role, b in role_bytes.items():
            us = prof["us"].get(role)
            if us is None:
                continue
            per.append({"name": role, "kernel": prof["kernel"].get(role), "us": us, "algorithmic_bytes": b, "GBps": b / us / 1e3, "frac": b / us / 1e3 / HBM_PEAK_GBPS})
            if "+attention" in role:
                per[-1]["note"] = ("one launch = the mat-vec AND the layer's whole attention (RoPE, KV append, scores, soft_max, V*P, Q4_0 of the result) behind "
                                   "in-launch hand-offs; only the mat-vec's weight bytes are counted, so this fraction is not comparable with the pure mat-vecs")
            mult = 1 if role == "output" else nl
            gb += b * mult; gu += us * mult
        dom = max(per, key=lambda p: p["algorithmic_bytes"] * (1 if p["name"] == "output" else nl))
        # HBM traffic of every launch, measured by the PMC passes of this run (kernel names are matched by their template prefix)
        if pmc:
            for p_ in per:
                kn = (p_["kernel"] or "")
                hit = [v for k, v in pmc.items() if kn and (k == kn or k.startswith(kn) or kn.startswith(k))]
                if hit:
                    p_["traffic_bytes"] = hit[0]
                    p_["traffic_over_algorithmic"] = hit[0] / p_["algorithmic_bytes"]
            traffic = dom.get("traffic_bytes")
            traffic_src = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE children of this run (separate passes, --kernel-trace only; 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md), 16 decode steps"
        # the launch that takes the most GPU time per token: k_qkv_attn -- the wq|wk|wv mat-vec AND the layer's attention; its bytes are
        # the mat-vec's plus the K and V rows it reads at the mean position of the profiled window (2 (t + 1) d fp32) and the two rows it appends
        by_time = max(per, key=lambda p: p["us"] * (1 if p["name"] == "output" else nl))
        t_prof = len(PROMPT) + max(args.warmup, 1) + min(args.steps, 96) // 2
        dbt = dict(by_time)
        if "+attention" in by_time["name"]:
            kvb = 2 * (t_prof + 1) * cfg["n_embd"] * 4 + 2 * cfg["n_embd"] * 4
            dbt.update({"kv_bytes_at_mean_position": kvb, "mean_position": t_prof, "algorithmic_bytes": by_time["algorithmic_bytes"] + kvb,
                        "GBps": (by_time["algorithmic_bytes"] + kvb) / by_time["us"] / 1e3, "frac": (by_time["algorithmic_bytes"] + kvb) / by_time["us"] / 1e3 / HBM_PEAK_GBPS})
            dbt.pop("note", None)
        dbt["share_of_gpu_time"] = by_time["us"] * (1 if by_time["name"] == "output" else nl) / prof["us"].get("token_span", float("nan"))
        roof["dominant_by_time"] = dbt
        roof.update({"kernel": f"{dom['kernel']} -- the {dom['name']} mat-vec as it runs in the captured decode step (norm prologue, SiLU*up -> Q4_0 epilogue); "
                               f"{dom['algorithmic_bytes'] * nl / gb * 100:.0f}% of the mat-vec bytes of a token",
                     "achieved": dom["GBps"], "frac": dom["frac"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "us_per_launch": dom["us"],
                     "traffic": traffic, "traffic_source": traffic_src if traffic else None,
                     "traffic_committed": traffic_committed, "traffic_error": None if traffic else pmc_err,
                     "method": f"rocprofv3 --kernel-trace of the same decode loop in a child process ({prof['tokens']} tokens); every dispatch labelled by its position in the "
                               "token's launch sequence; average kernel duration",
                     "in_situ_per_launch": per,
                     "other_launches_us": {k: v for k, v in prof["us"].items() if k not in role_bytes},
                     "all_gemv_launches_of_a_token": {"GBps": gb / gu / 1e3, "frac": gb / gu / 1e3 / HBM_PEAK_GBPS, "bytes": gb, "us": gu}})
    else:
        dom = max(r["shapes"], key=lambda s: s["algo_bytes"] * (1 if s["name"] == "output" else nl))
        roof.update({"kernel": "lh::k_gemv PRE_QA / STORE PROBE VARIANT on w1|w3 (the in-situ profile was unavailable: " + str(prof_err) + ")",
                     "achieved": dom["GBps"], "frac": dom["GBps"] / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": dom["algo_bytes"],
                     "us_per_launch": dom["us_per_launch"], "traffic": None, "traffic_committed": traffic_committed, "method": "HIP events over back-to-back launches (probe variant, not in situ)"})

    result = {
        "metric": "decode tokens/sec LLaMA-7B Q4_0 @1 GPU; % HBM-roofline on Q4_0 GEMV",
        "value": tps, "unit": "tokens/s", "n_gpus": 1, "steps": r["steps"], "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "q4_0 x q4_0 -> int32 block sums, fp32 scales/accumulate",
        "data": "synthetic (random-init 7B-architecture weights in the reference file format, synthetic token ids)",
        "config": {"workload": f"LLaMA-{args.model} Q4_0 single-token decode, greedy, n_ctx {args.n_ctx}: `value` = {r['steps']} timed tokens at context positions "
                               f"{r['n_past0']} .. {r['n_past0'] + r['steps'] - 1} (after 8 prompt + {args.warmup} warm-up tokens); BASELINE.json configs[1] proper -- the generation "
                               f"to the end of the {args.n_ctx}-token context -- is config.full_context_tokens_per_s / roofline.full_context_*",
                   "n_threads_semantics": args.threads, "parallelism": "1 GPU"},
        "value_pcie": r["value_pcie"],
        "load_s": r["t_load"],
        "roofline": roof,
        "parity": parity,
    }
    if r.get("full"):
        fc = r["full"]
        fb = token_bytes(cfg, len(PROMPT) + args.warmup + fc["steps"] // 2)
        fms = fc["seconds"] * 1e3 / fc["steps"]
        result["full_context"] = {"workload": f"configs[1]: {fc['steps']} generated tokens to context {args.n_ctx} (8 prompt + {args.warmup} warm-up before them)",
                                  "steps": fc["steps"], "tokens_per_s": fc["steps"] / fc["seconds"], "ms_per_step": fms,
                                  "end_to_end_frac": fb / (fms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "tokens_continue_the_timed_run": fc["same_prefix"]}
    elif r["steps"] >= args.n_ctx - len(PROMPT) - args.warmup:
        result["full_context"] = {"workload": "the timed run IS configs[1]: it fills the context", "steps": r["steps"], "tokens_per_s": tps, "ms_per_step": ms_step,
                                  "end_to_end_frac": roof["end_to_end_frac"]}
    if cpu_base is not None:
        result["cpu_baseline"] = cpu_base
    result["prefill"] = r["prefill"]
    if not args.no_prefill_2048 and args.model in ("7B", "13B"):
        try:
            result["prefill"]["configs2_2048_tokens_one_eval"] = prefill_2048(args, cfg, path)
        except Exception as e:
            result["prefill"]["configs2_2048_tokens_one_eval"] = {"error": repr(e)}
    if args.model == "7B" and not args.no_concurrent:
        try:
            result["concurrent_sequences"] = [concurrent_sequences(args, cfg, path, n) for n in (2, 4)]
        except Exception as e:
            result["concurrent_sequences"] = {"error": repr(e)}
    if args.model in ("7B", "13B") and not args.no_concurrent:
        try:
            result["batched_sequences"] = [batched_sequences(args, cfg, path, n) for n in (2, 4, 8)]
        except Exception as e:
            result["batched_sequences"] = {"error": repr(e)}
    # the layer pipeline behind the C ABI (one llamahip_model_load with a device list, include/llamahip.h): every stage on THIS GPU -- the
    # stage launches, streams, events and hand-off copies of the multi-GPU handle, none of its parallel hardware (secondary figure)
    if args.model == "7B" and not args.no_concurrent:
        try:
            result["inprocess_pipeline"] = [inprocess_pipeline(args, cfg, path, n, r["gpu_trace"]) for n in (2, 8)]
        except Exception as e:
            result["inprocess_pipeline"] = {"error": repr(e)}
        try:        # ... and several sequences through it (llamahip_decode_greedy_multi): one stage x 16, two stages x 32, four stages x 32 sequences
            result["inprocess_pipeline_multi"] = [inprocess_pipeline_multi(args, cfg, path, st, sq, r["gpu_trace"]) for st, sq in ((1, 16), (2, 32), (4, 32))]
        except Exception as e:
            result["inprocess_pipeline_multi"] = {"error": repr(e)}
    # figures that must survive a reader that keeps only metric / value / config / roofline of this line
    if "full_context" in result:
        result["roofline"]["full_context_tokens_per_s"] = result["full_context"]["tokens_per_s"]
        result["roofline"]["full_context_end_to_end_frac"] = result["full_context"]["end_to_end_frac"]
        result["config"]["full_context_tokens_per_s"] = result["full_context"]["tokens_per_s"]
    # (scalars only: nested objects of config / roofline do not survive such a reader)
    if isinstance(result.get("inprocess_pipeline"), list):
        for b in result["inprocess_pipeline"]:
            result["config"][f"inprocess_pipeline_{b['stages']}_stages_one_gpu_tokens_per_s"] = round(b["tokens_per_s"], 1)
        result["config"]["inprocess_pipeline_tokens_equal_single_device"] = all(b["tokens_equal_single_device"] for b in result["inprocess_pipeline"])
    if isinstance(result.get("inprocess_pipeline_multi"), list):
        for b in result["inprocess_pipeline_multi"]:
            result["config"][f"inprocess_multi_{b['stages']}_stages_{b['sequences']}_seq_aggregate_tokens_per_s"] = round(b["aggregate_tokens_per_s"], 1)
        result["config"]["inprocess_multi_sequence0_equal_single_stream"] = all(b["sequence0_tokens_equal_single_stream"] for b in result["inprocess_pipeline_multi"])
    if isinstance(result.get("batched_sequences"), list):
        for b in result["batched_sequences"]:
            result["config"][f"batched_sequences_aggregate_tokens_per_s_{b['sequences']}_seq"] = round(b["aggregate_tokens_per_s"], 1)
            result["config"][f"batched_sequences_ms_per_step_{b['sequences']}_seq"] = round(b["ms_per_step"], 4)
        result["config"]["batched_sequences_tokens_equal_single_stream"] = all(b["tokens_equal_single_stream"] for b in result["batched_sequences"])
    result["config"]["parity_checked"] = bool(parity.get("checked"))
    result["config"]["parity_identical"] = parity.get("identical")
    result["config"]["parity_tokens_compared"] = parity.get("tokens_compared")
    result["config"]["parity_against"] = parity.get("against")
    dbt_ = result["roofline"].get("dominant_by_time")
    if isinstance(dbt_, dict):
        result["roofline"]["dominant_by_time_kernel"] = dbt_.get("kernel")
        result["roofline"]["dominant_by_time_us"] = dbt_.get("us")
        result["roofline"]["dominant_by_time_frac"] = dbt_.get("frac")
        result["roofline"]["frac_dominant_by_time"] = dbt_.get("frac")          # (next to `frac`: the launch with the most GPU time, K / V rows counted)
        result["roofline"]["dominant_by_time_share"] = dbt_.get("share_of_gpu_time")
    p9 = result["prefill"].get("reference_9_token_chunks") if isinstance(result.get("prefill"), dict) else None
    if isinstance(p9, dict):
        result["config"]["reference_9_token_evals_tokens_per_s"] = round(p9["tokens_per_s"], 1)
    p2k = result["prefill"].get("configs2_2048_tokens_one_eval") if isinstance(result.get("prefill"), dict) else None
    if isinstance(p2k, dict) and "tokens_per_s" in p2k:
        result["config"]["prefill_2048_tokens_per_s"] = round(p2k["tokens_per_s"], 1)
    # the reference's user-facing flow (LlamaRunner.run: load once, 8-token prompt batches, one llama_eval and one
    # host-side top-k / top-p sample per token, token text through the event callback) -- not the headline metric
    try:
        from llama_swift_amd import Config, LlamaRunner
        rng = np.random.default_rng(7)
        stamps = []
        rn = LlamaRunner(path)
        cfgr = Config(numThreads=args.threads, numTokens=320, n_ctx=args.n_ctx, keepModel=True)
        prompt_text = "".join("tok%05d" % t for t in rng.integers(3, cfg["n_vocab"], 16))
        rn.run(prompt_text, cfgr)                                             # loads the model, warms up
        rn.run(prompt_text, cfgr, tokenHandler=lambda _t: stamps.append(time.perf_counter()))
        rn.close()
        if len(stamps) > 64:
            gen = stamps[-257:] if len(stamps) >= 257 + 17 else stamps[17:]   # generated tokens only (the prompt is echoed first)
            result["runner"] = {"sampled_tokens_per_s": (len(gen) - 1) / (gen[-1] - gen[0]), "tokens": len(gen) - 1,
                                "note": "LlamaRunner.run with its default sampling (top_k 40, top_p 0.95, temp 0.8, repeat penalty 1.3), model kept from a previous run"}
    except Exception as e:                                                    # the runner leg must never cost the headline line
        log(f"[bench] runner leg skipped: {e}")
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

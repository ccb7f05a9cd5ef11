"""Layer-pipelined decode across the GPUs of one node.

The residual stream ``inpL`` (fp32 ``[N, n_embd]``) is the only tensor that crosses layers
(LlamaPredictOperation.mm:563-564, 687-690) and the KV cache is indexed by layer (.mm:586-587), so
the model shards by contiguous layer ranges with ONE exchange per stage boundary: a point-to-point
send/recv of ``N * n_embd * 4`` bytes (32 KiB per LLaMA-65B decode token) -- no all-reduce anywhere.
One process per GPU; ``torch.distributed`` is the plumbing (backend ``nccl`` = RCCL over xGMI on the
GPU box, ``gloo`` in the CPU tests).  The last stage picks the token (greedy) and returns its id to
stage 0.

A single greedy stream is strictly sequential through the stages, so throughput comes from keeping
``n_seq`` independent sequences in flight (each stage holds one KV cache per sequence).  All sends
are non-blocking (``isend``): with blocking sends the token feedback edge closes a cycle of
rendezvous and the pipeline deadlocks once every stage holds an item.  On NCCL/RCCL, non-blocking is
not enough: all point-to-point operations of one communicator execute in issue order on one internal
stream, so stage 0's S-th hand-off (queued before its first token receive) would wait for the last
stage, whose token send waits for exactly that receive.  The token feedback therefore travels on its
OWN process group (own communicator, own stream); the forward edges alone form a DAG.

The schedule is independent of what a *stage* is: anything with ``run(seq, n_past, tokens, hidden)``.
The product stage is :class:`HipStage` (C ABI ``llamahip_eval_stage``); the CPU tests plug in an
oracle-backed stage to check the schedule itself.
"""
from __future__ import annotations

import json
import os
import time
from typing import Optional, Protocol, Sequence

import numpy as np


class Stage(Protocol):
    is_first: bool
    is_last: bool
    n_embd: int
    n_vocab: int
    device: str

    def run(self, seq: int, n_past: int, tokens: Optional[np.ndarray], hidden):
        """first stage: tokens -> hidden ; middle: hidden -> hidden ; last: ... -> logits (np.ndarray).
        hidden is a torch tensor [N * n_embd] fp32 on ``device``."""
        ...


def layer_range(n_layer: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, as even as possible; earlier stages take the remainder."""
    base, rem = divmod(n_layer, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipStage:
    """One pipeline stage on one MI355X through the C ABI (no CPU fallback)."""

    def __init__(self, path: str, n_ctx: int, rank: int, world: int, n_seq: int, n_layer: int, device_index: int,
                 n_threads: int = 8):
        import torch

        from . import binding
        lo, hi = layer_range(n_layer, rank, world)
        self.model = binding.Model(path, n_ctx=n_ctx, device=device_index, layer_begin=lo, layer_end=hi, n_seq=n_seq)
        self.is_first, self.is_last = lo == 0, hi == n_layer
        self.n_embd, self.n_vocab = self.model.n_embd, self.model.n_vocab
        self.device = f"cuda:{device_index}"
        self.n_threads = n_threads
        self._torch = torch

    def run(self, seq, n_past, tokens, hidden):
        torch = self._torch
        self.model.set_seq(seq)
        n = len(tokens) if tokens is not None else hidden.numel() // self.n_embd
        out = None if self.is_last else torch.empty(n * self.n_embd, dtype=torch.float32, device=self.device)
        logits = self.model.eval_stage(
            n_past, tokens=tokens if self.is_first else None, n_tokens=n,
            hidden_in=0 if self.is_first else hidden.data_ptr(), hidden_out=0 if self.is_last else out.data_ptr(),
            want_logits=self.is_last, n_threads=self.n_threads)
        return logits if self.is_last else out


def pipeline_rounds(stage: Stage, rank: int, world: int, dist, torch, tokens_per_seq: Sequence[np.ndarray],
                    n_past: Sequence[int], rounds: int, token_group=None):
    """Runs `rounds` pipeline rounds.  Round 0 feeds tokens_per_seq[s] (a prompt chunk or one token) for
    every sequence s; every later round feeds the token the last stage picked in the previous round.
    Returns (tokens [n_seq][rounds] on every rank, final n_past list)."""
    S = len(tokens_per_seq)
    dev = stage.device
    n_past = list(n_past)
    picked = np.zeros((S, rounds), np.int32)
    pending = []                                 # in-flight isend handles (+ the tensors they read)
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    cur = [np.ascontiguousarray(t, np.int32) for t in tokens_per_seq]

    def reap(limit):
        while len(pending) > limit:
            w, _keep = pending.pop(0)
            w.wait()

    for k in range(rounds):
        for s in range(S):
            n = len(cur[s]) if (k == 0) else 1
            if stage.is_first:
                if k > 0 and world > 1:          # token picked by the last stage for (k-1, s)
                    t = torch.zeros(1, dtype=torch.int32, device=dev)
                    dist.recv(t, src=world - 1, group=token_group)
                    cur[s] = t.cpu().numpy().astype(np.int32)
                    picked[s, k - 1] = int(cur[s][0])
                hidden = None
            else:
                hidden = torch.empty(n * stage.n_embd, dtype=torch.float32, device=dev)
                dist.recv(hidden, src=prv)
            if dev.startswith("cuda"):
                torch.cuda.current_stream().synchronize()        # recv is stream-ordered; the C ABI uses its own stream
            out = stage.run(s, n_past[s], cur[s] if stage.is_first else None, hidden)
            n_past[s] += n
            if stage.is_last:
                tok = int(np.argmax(out))                          # greedy: lowest index on ties
                picked[s, k] = tok
                if world > 1:
                    t = torch.tensor([tok], dtype=torch.int32, device=dev)
                    pending.append((dist.isend(t, dst=0, group=token_group), t))
                else:
                    cur[s] = np.array([tok], np.int32)
            else:
                pending.append((dist.isend(out, dst=nxt), out))
            reap(2 * S)
    # drain: the first stage still has the last round's tokens to receive
    if stage.is_first and world > 1:
        for s in range(S):
            t = torch.zeros(1, dtype=torch.int32, device=dev)
            dist.recv(t, src=world - 1, group=token_group)
            picked[s, rounds - 1] = int(t.cpu()[0])
    reap(0)
    if world > 1:                                # every rank reports the same token matrix
        buf = torch.from_numpy(picked).to(dev)
        dist.broadcast(buf, src=0)
        picked = buf.cpu().numpy()
    return picked, n_past


def bench_main(args, cfg, model_path_fn, log):
    """`bench.py --gpus N` for N > 1 (launched by torch.distributed.run, one rank per GPU)."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} must be launched with {args.gpus} ranks (WORLD_SIZE={world}); "
                         f"use: python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    token_group = dist.new_group(list(range(world)))          # separate communicator for the feedback edge
    if rank == 0:
        path = model_path_fn(args.model, cfg, args.seed)
    dist.barrier()
    path = model_path_fn(args.model, cfg, args.seed)

    S = world                                    # one sequence in flight per stage (weak scaling)
    stage = HipStage(path, args.n_ctx, rank, world, S, cfg["n_layer"], local, args.threads)
    rng = np.random.default_rng(1234)
    prompts = [np.concatenate([[1], rng.integers(3, cfg["n_vocab"], 7)]).astype(np.int32) for _ in range(S)]
    steps = min(args.steps, args.n_ctx - 8 - args.warmup - 1)
    # prompt round (8 tokens per sequence) + warm-up rounds, untimed
    toks, n_past = pipeline_rounds(stage, rank, world, dist, torch, prompts, [0] * S, 1 + args.warmup, token_group)
    last = [np.array([toks[s, -1]], np.int32) for s in range(S)]
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks2, n_past = pipeline_rounds(stage, rank, world, dist, torch, last, n_past, steps, token_group)
    dist.barrier(); torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=f"cuda:{local}")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    if rank == 0:
        total = S * steps
        print(json.dumps({
            "metric": "decode tokens/sec LLaMA-7B Q4_0 @1 GPU; % HBM-roofline on Q4_0 GEMV",
            "value": total / dt, "unit": "tokens/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "q4_0 x q4_0 -> int32 block sums, fp32 scales/accumulate",
            "data": "synthetic (random-init weights in the reference file format, synthetic token ids)",
            "config": {"workload": f"LLaMA-{args.model} Q4_0 greedy decode, {world}-stage layer pipeline "
                                   f"({cfg['n_layer']} layers / {world}), {S} independent sequences in flight, n_ctx {args.n_ctx}; "
                                   f"a step = one token for every sequence",
                       "parallelism": f"pp{world} (RCCL p2p hand-off of the fp32 residual stream)",
                       "sequences": S, "tokens_timed": total},
            "single_stream_tokens_per_s_estimate": steps / dt,
        }), flush=True)
    dist.destroy_process_group()

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

Two schedules share these rules:

* :func:`pipeline_rounds` -- host-synchronous, any number of tokens per item (prompt chunks).  Every
  hand-off returns to the host (``llamahip_eval_stage``).
* :func:`pipeline_decode` -- the steady-state decode loop, fully stream-ordered: receive -> stage step
  (one hipGraph launch, ``llamahip_stage_step``) -> send are enqueued on the device and the host never
  waits; the last stage picks the token on the device and the position advances on the device.  The
  forward edges alternate between two process groups by parity of the sending rank, so a rank's
  receive (from r-1) and its send (to r+1) never share a communicator stream and the next item's
  receive can be posted while the previous send is still in flight.

The schedules are independent of what a *stage* is.  The product stage is :class:`HipStage` (C ABI);
the CPU tests plug in an oracle-backed stage to check the schedules themselves under ``gloo``.
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

    # --- stream-ordered single-token steps (pipeline_decode) ---
    tok_in: list      # per sequence: int32[1] tensor on ``device`` (first stage reads it)
    tok_out: list     # per sequence: int32[1] tensor (last stage writes the greedy pick)
    hid_in: list      # per sequence: fp32[n_embd] tensor (None on the first stage)
    hid_out: list     # per sequence: fp32[n_embd] tensor (None on the last stage)

    def bind(self, seq: int, n_past: int, first_token: int) -> None:
        """Position of the next token of `seq` and (first stage) the token itself; allocates the i/o tensors."""
        ...

    def step(self, seq: int) -> None:
        """One token through this stage's layers: tok_in|hid_in -> hid_out|tok_out, asynchronous on
        the device's current stream; the position advances by one."""
        ...

    def trace(self, seq: int, cap: int):
        """Waits for the device; (steps since bind, position, tokens picked [last stage only])."""
        ...


ERR_PREDICT = -1001          # LlamaErrorCodePredictionFailed (Sources/llamaObjCxx/headers/LlamaError.h:18)


class PipelineError(RuntimeError):
    """A stage lost its peer (or the schedule stopped making progress): what the bridge reports as PredictionFailed."""

    def __init__(self, message: str):
        super().__init__(f"[com.alexrozanski.llama.error {ERR_PREDICT}] {message}")
        self.code, self.message = ERR_PREDICT, message


def run_guarded(fn, rank: int, world: int, limit_s: float, what: str, on_timeout=None):
    """Runs one pipeline schedule call with the two failure paths a multi-rank decode has:
      * a peer that went away surfaces as an exception of the transport (gloo: connection reset; RCCL: an async error
        on the work handle) -> re-raised as PipelineError (code -1001, message names the rank and the call);
      * a peer that hangs (or an RCCL receive that never matches) surfaces as NOTHING -> a watchdog ends the process
        after `limit_s` seconds without the call returning: message on stderr, exit code 3 (a blocked collective cannot
        be interrupted from Python, and a stuck rank must not keep its GPU and its launcher forever).
    `on_timeout` (tests) replaces the process exit."""
    import threading

    def _expired():
        msg = f"[pipeline] rank {rank}/{world}: no progress for {limit_s:.0f} s in {what} -- PredictionFailed ({ERR_PREDICT}), leaving"
        os.write(2, (msg + "\n").encode())
        if on_timeout:
            on_timeout(msg)
        else:
            os._exit(3)

    timer = threading.Timer(limit_s, _expired)
    timer.daemon = True
    timer.start()
    try:
        return fn()
    except PipelineError:
        raise
    except Exception as e:                       # transport errors are RuntimeError / DistBackendError, worded by the backend
        raise PipelineError(f"rank {rank}/{world}: {what} failed: {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}") from e
    finally:
        timer.cancel()


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

    # --- device-side mailboxes (no collective per token) ---
    def setup_mailboxes(self, dist, rank: int, world: int, n_seq: int) -> None:
        """Bootstrap, once: every stage creates its inboxes, the 64-byte IPC handles travel over the process group (ONE object
        all-gather), every stage opens its successor's hidden inbox and the last stage the first stage's token inbox.  From then on
        a token step is `step(seq)` alone: the row and the token move between the GPUs inside the kernels (include/llamahip.h)."""
        mine = []
        for s in range(n_seq):
            _, _, hh, th = self.model.stage_mailbox(s)
            mine.append((hh, th))
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        for s in range(n_seq):
            nxt_h = everyone[rank + 1][s][0] if rank + 1 < world else None
            tok_h = everyone[0][s][1] if (rank == world - 1 and world > 1) else None
            self.model.stage_mailbox_connect(s, next_hidden_handle=nxt_h, token_handle=tok_h)
        self.mailboxes = True

    # --- stream-ordered steps ---
    def bind(self, seq, n_past, first_token):
        torch = self._torch
        if getattr(self, "mailboxes", False):
            if not hasattr(self, "tok_in"):
                self.tok_in = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(self.model.n_seq)]
            self.tok_in[seq].fill_(int(first_token))
            torch.cuda.current_stream().synchronize()
            self.model.stage_bind(seq, n_past, token_in=self.tok_in[seq].data_ptr() if self.is_first else 0)
            return
        if not hasattr(self, "tok_in"):
            S, dev = self.model.n_seq, self.device
            self.tok_in = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(S)]
            # a whole-model stage feeds its own pick back: same buffer
            self.tok_out = self.tok_in if (self.is_first and self.is_last) else [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(S)]
            self.hid_in = [None if self.is_first else torch.zeros(self.n_embd, dtype=torch.float32, device=dev) for _ in range(S)]
            self.hid_out = [None if self.is_last else torch.zeros(self.n_embd, dtype=torch.float32, device=dev) for _ in range(S)]
        self.tok_in[seq].fill_(int(first_token))
        torch.cuda.current_stream().synchronize()
        ptr = lambda t: 0 if t is None else t.data_ptr()
        self.model.stage_bind(seq, n_past,
                              token_in=ptr(self.tok_in[seq]) if self.is_first else 0,
                              hidden_in=ptr(self.hid_in[seq]), hidden_out=ptr(self.hid_out[seq]),
                              token_out=ptr(self.tok_out[seq]) if self.is_last else 0)

    def step(self, seq):
        self.model.stage_step(seq, self.n_threads, self._torch.cuda.current_stream().cuda_stream)

    def trace(self, seq, cap):
        return self.model.stage_trace(seq, cap)


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


def pipeline_decode(stage: Stage, rank: int, world: int, dist, n_seq: int, rounds: int,
                    fwd_groups=None, token_group=None):
    """`rounds` greedy tokens for each of `n_seq` bound sequences (stage.bind), one token per sequence
    per round, with no host synchronisation: every receive, stage step and send is enqueued in program
    order and ordered on the device.  Round 0 evaluates the token already in ``stage.tok_in`` (from
    bind or from the previous call); the call ends with stage 0 receiving (stream-ordered) the last
    round's picks into ``tok_in``, so calls can be chained and no send is left unmatched.
    Returns nothing: read the picks with ``stage.trace`` on the last stage."""
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    grp = (lambda sender: fwd_groups[sender % 2]) if fwd_groups else (lambda sender: None)
    pending = []

    def reap(limit):
        while len(pending) > limit:
            pending.pop(0).wait()

    for k in range(rounds):
        for s in range(n_seq):
            if stage.is_first:
                if world > 1 and k > 0:
                    dist.recv(stage.tok_in[s], src=world - 1, group=token_group)     # pick of the previous round
            else:
                dist.recv(stage.hid_in[s], src=prv, group=grp(prv))
            stage.step(s)
            if not stage.is_last:
                pending.append(dist.isend(stage.hid_out[s], dst=nxt, group=grp(rank)))
            elif world > 1:
                pending.append(dist.isend(stage.tok_out[s], dst=0, group=token_group))
            reap(2 * n_seq)
    if stage.is_first and world > 1 and rounds > 0:
        for s in range(n_seq):
            dist.recv(stage.tok_in[s], src=world - 1, group=token_group)
    reap(0)


def mailbox_decode(stage: Stage, n_seq: int, rounds: int, seqs: Optional[Sequence[int]] = None) -> None:
    """The decode loop with device-side mailboxes: `rounds` token steps for each sequence, enqueued back to back on the current
    stream.  No receive, no send, no ordering with the other ranks on the host or on a communicator: a stage's first kernel polls
    its inbox, its last kernel stores into the neighbour's (every poll is bounded; a lost neighbour surfaces from stage.trace)."""
    for _ in range(rounds):
        for s in (seqs if seqs is not None else range(n_seq)):
            stage.step(s)


def gather_traces(stage: Stage, rank: int, world: int, dist, torch, n_seq: int, cap: int):
    """Tokens picked since bind, [n_seq][cap] on every rank (the last stage owns them)."""
    out = np.zeros((n_seq, cap), np.int32)
    pos = [0] * n_seq
    for s in range(n_seq):
        n, pos[s], toks = stage.trace(s, cap)
        if stage.is_last:
            out[s, :len(toks)] = toks
    if world > 1:
        buf = torch.from_numpy(out).to(stage.device)
        dist.broadcast(buf, src=world - 1)
        out = buf.cpu().numpy()
    return out, pos


def _cpu_trace(path: str, prompt: np.ndarray, n_tokens: int, n_ctx: int, budget_s: float):
    """Greedy tokens of the CPU path (the reference's ggml.c build when it travelled with the snapshot, else the restatement) for one
    prompt: the parity gate of the multi-GPU line.  Bounded by `budget_s` seconds of decoding."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
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


def bench_main(args, cfg, model_path_fn, log, models=None):
    """`bench.py --gpus N` for N > 1 (launched by torch.distributed.run, one rank per GPU)."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} must be launched with {args.gpus} ranks (WORLD_SIZE={world}); "
                         f"use: python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}")
    # RCCL prints a version banner on STDOUT when it creates a communicator; the bench contract is ONE
    # JSON line on stdout, so everything before that line goes to stderr at the file-descriptor level
    import sys
    import threading
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    # a multi-rank run that stops making progress (a peer died, a hand-off never matched) must not hang the
    # caller forever: every schedule call below runs under run_guarded (transport error -> PipelineError, silence ->
    # exit code 3 after `limit` seconds), and the whole bench under one more timer of the same length
    limit = float(os.environ.get("LLAMAHIP_PIPE_WATCHDOG_S", "900"))
    headline = {}                                # rank 0: the finished JSON line of the headline model (printed by whoever ends the run)

    def _emit_and_exit(code):
        if rank == 0 and headline:
            os.dup2(saved_stdout, 1)
            os.write(1, (json.dumps(headline) + "\n").encode())
        os._exit(code)

    def _abort():
        os.write(2, f"[bench] rank {rank}/{world}: no result after {limit:.0f} s -- PredictionFailed ({ERR_PREDICT}), aborting\n".encode())
        _emit_and_exit(0 if headline else 3)     # (a stuck EXTRA leg must not cost the headline line that is already measured)

    watchdog = threading.Timer(limit, _abort)
    watchdog.daemon = True
    watchdog.start()
    guard = lambda fn, what: run_guarded(fn, rank, world, limit, what, on_timeout=(lambda msg: _emit_and_exit(0)) if headline else None)
    # (smoke test of the multi-rank path on ONE GPU: LLAMAHIP_PIPE_ONE_GPU=1 puts every rank on cuda:0 and LLAMAHIP_PIPE_BACKEND=gloo
    #  replaces RCCL, which refuses two ranks on one device; the mailboxes then run over HIP IPC between the processes)
    if os.environ.get("LLAMAHIP_PIPE_ONE_GPU") == "1":
        local = 0
    backend = os.environ.get("LLAMAHIP_PIPE_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    else:
        dist.init_process_group(backend)
    token_group = dist.new_group(list(range(world)))          # separate communicator for the feedback edge
    fwd_groups = [dist.new_group(list(range(world))), dist.new_group(list(range(world)))]   # forward edges by sender parity
    sync_schedule = os.environ.get("LLAMAHIP_PIPELINE_SYNC", "0") == "1"
    want_mailbox = os.environ.get("LLAMAHIP_PIPE_MAILBOX", "1") != "0" and not sync_schedule and world > 1

    def run_model(model_name, mcfg, steps_req, warmup, parity_tokens):
        if rank == 0:
            model_path_fn(model_name, mcfg, args.seed)
        dist.barrier()
        path = model_path_fn(model_name, mcfg, args.seed)
        # sequences in flight: two per stage (weak scaling).  With exactly one per stage every stage waits out
        # the hand-off latency of its predecessor on every step; a second one keeps a ready item queued.
        S = world * max(1, int(os.environ.get("LLAMAHIP_PIPE_SEQS_PER_STAGE", "2"))) if world > 1 else 1
        stage = HipStage(path, args.n_ctx, rank, world, S, mcfg["n_layer"], local, args.threads)
        rng = np.random.default_rng(1234)
        prompts = [np.concatenate([[1], rng.integers(3, mcfg["n_vocab"], 7)]).astype(np.int32) for _ in range(S)]
        n_single = 16                                          # single-stream latency leg: tokens of sequence 0 alone
        steps = max(1, min(steps_req, args.n_ctx - 8 - warmup - 1 - n_single))
        hand_off = "RCCL point-to-point per token (torch.distributed isend / recv, stream-ordered)"
        if sync_schedule:
            toks, n_past = guard(lambda: pipeline_rounds(stage, rank, world, dist, torch, prompts, [0] * S, 1 + warmup, token_group), "pipeline_rounds (prompt + warm-up)")
            last = [np.array([toks[s, -1]], np.int32) for s in range(S)]
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            toks2, n_past = guard(lambda: pipeline_rounds(stage, rank, world, dist, torch, last, n_past, steps, token_group), "pipeline_rounds (timed decode)")
            dist.barrier(); torch.cuda.synchronize()
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
                flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local}")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                mailbox = int(flag.item()) == 1
                if not mailbox:
                    stage.mailboxes = False
            for s in range(S):
                stage.bind(s, n_past[s], firsts[s])
            lane = torch.cuda.Stream()               # the decode loop's own stream
            if mailbox:
                # handshake: ONE token of sequence 0 through every stage on the mailboxes, then every rank reads its fault word.  A
                # row that does not arrive (peer mapping that does not carry stores, ...) costs one poll bound here, not one per step
                # of the timed loop; all ranks agree on the outcome and fall back to the RCCL hand-off together.
                ok = 1
                try:
                    with torch.cuda.stream(lane):
                        stage.step(0)
                    torch.cuda.synchronize()
                    n_done, pos0, _ = stage.trace(0, 1)
                    ok = int(n_done == 1 and pos0 == n_past[0] + 1)
                except Exception as e:
                    log(f"[bench] rank {rank}: mailbox handshake failed ({type(e).__name__}: {str(e)[:200]}); RCCL hand-off")
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local}")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) != 1:
                    mailbox = False
                    stage.mailboxes = False
                    toks, n_past = guard(lambda: pipeline_rounds(stage, rank, world, dist, torch, prompts, [0] * S, 1, token_group), "pipeline_rounds (prompt, again)")
                    firsts = [int(toks[s, -1]) for s in range(S)]
                    for s in range(S):
                        stage.bind(s, n_past[s], firsts[s])
            handshake_tokens = 1 if mailbox else 0
            if mailbox:
                hand_off = "device-side mailboxes: position-tagged granules stored into the next stage's memory (HIP IPC / xGMI) by the last kernel of a stage step, polled by the first kernel of the next; no collective and no host call per token"

            def decode(n, seqs=None):
                with torch.cuda.stream(lane):
                    if mailbox:
                        mailbox_decode(stage, S, n, seqs)
                    else:
                        pipeline_decode(stage, rank, world, dist, S if seqs is None else len(seqs), n, fwd_groups, token_group)
                torch.cuda.synchronize()
            guard(lambda: decode(warmup), "decode (warm-up)")                                # untimed; captures the graphs
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            guard(lambda: decode(steps), "decode (timed)")
            dist.barrier(); torch.cuda.synchronize()
            dt_loc = time.perf_counter() - t0
            # single-stream latency, measured: sequence 0 alone through all stages
            single = None
            if world > 1:
                dist.barrier(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                guard(lambda: decode(n_single, [0]), "decode (single stream)")
                dist.barrier(); torch.cuda.synchronize()
                ds = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=f"cuda:{local}")
                dist.all_reduce(ds, op=dist.ReduceOp.MAX)
                single = {"tokens": n_single, "ms_per_token": float(ds.item()) * 1e3 / n_single, "tokens_per_s": n_single / float(ds.item()),
                          "note": "sequence 0 alone: one token at a time through every stage (the latency a single user sees)"}
            traces, _pos = guard(lambda: gather_traces(stage, rank, world, dist, torch, S, handshake_tokens + warmup + steps + (n_single if world > 1 else 0)), "gather_traces")
        dt = torch.tensor([dt_loc], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dt = float(dt.item())
        # parity gate: sequence 0's prompt pick and first generated tokens against the CPU path (rank 0 computes it, bounded)
        parity = {"checked": False}
        if rank == 0 and parity_tokens > 0:
            try:
                kind, cfirst, ctoks = _cpu_trace(path, prompts[0], parity_tokens, args.n_ctx, float(os.environ.get("LLAMAHIP_PIPE_PARITY_S", "30")))
                got = [int(t) for t in traces[0][:len(ctoks)]]
                parity = {"checked": True, "against": f"{kind} CPU path, 8 threads, sequence 0", "prompt_pick_identical": cfirst == firsts[0],
                          "tokens_compared": len(ctoks), "identical": cfirst == firsts[0] and got == ctoks,
                          "first_divergence": next((i for i, (x, y) in enumerate(zip(got, ctoks)) if x != y), None)}
            except Exception as e:                              # the checker must never take the measurement down
                parity = {"checked": False, "error": repr(e)}
        roof = None
        try:
            r = stage.model.bench_gemv(2, -1, 1, 10)
            roof = {"bound": "hbm", "kernel": "lh::k_gemv PRE_QA / STORE probe variant on w1|w3 of rank 0's layers (stand-alone, not in situ)", "achieved": r["GBps"], "peak": 8000.0,
                    "unit": "GB/s", "frac": r["GBps"] / 8000.0, "traffic": None, "algorithmic_bytes_per_launch": r["algo_bytes"], "us_per_launch": r["us_per_launch"],
                    "per_stage_weight_bytes": stage.model.stats()["weight_bytes_device"]}
        except Exception as e:                       # measurement extras never cost the headline line
            roof = {"error": repr(e)}
        stage.model.close()
        return dict(S=S, steps=steps, dt=dt, parity=parity, roof=roof, single=single, hand_off=hand_off, n_layer=mcfg["n_layer"])

    r = run_model(args.model, cfg, args.steps, args.warmup, 8)
    if rank == 0:
        total = r["S"] * r["steps"]
        headline.update({
            "metric": f"decode tokens/sec LLaMA-{args.model} Q4_0 @{world} GPUs (layer pipeline, {r['S']} sequences in flight); % HBM-roofline on Q4_0 GEMV",
            "value": total / r["dt"], "unit": "tokens/s", "n_gpus": world, "steps": r["steps"], "warmup": args.warmup,
            "ms_per_step": r["dt"] * 1e3 / r["steps"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "q4_0 x q4_0 -> int32 block sums, fp32 scales/accumulate",
            "data": "synthetic (random-init weights in the reference file format, synthetic token ids)",
            "config": {"workload": f"LLaMA-{args.model} Q4_0 greedy decode, {world}-stage layer pipeline "
                                   f"({r['n_layer']} layers / {world}), {r['S']} independent sequences in flight, n_ctx {args.n_ctx}; "
                                   f"a step = one token for every sequence",
                       "parallelism": f"pp{world}", "hand_off": r["hand_off"],
                       "sequences": r["S"], "tokens_timed": total},
            "roofline": r["roof"],
            "parity": r["parity"],
            "cpu_baseline": None,
            "cpu_baseline_note": "timed at N = 1 only (bench.py --gpus 1)",
            "single_stream": r["single"],
            "schedule": "host-synchronous" if sync_schedule else "stream-ordered (hipGraph stage steps, device-side greedy pick)",
        })
    # BASELINE.json configs[4]: the 65B model is what the 8-GPU pipeline is for.  A bounded extra leg (32 timed steps), reported next to
    # the headline; whatever happens to it, the headline line above is printed.
    if models and args.model != "65B" and os.environ.get("LLAMAHIP_BENCH_65B", "1") != "0" and "65B" in models:
        try:
            r65 = run_model("65B", models["65B"], 32, 4, 4)
            if rank == 0:
                t65 = r65["S"] * r65["steps"]
                headline["config4_65B"] = {"workload": f"LLaMA-65B Q4_0, {r65['n_layer']} layers over {world} stages, {r65['S']} sequences in flight",
                                           "tokens_per_s": t65 / r65["dt"], "ms_per_step": r65["dt"] * 1e3 / r65["steps"], "steps": r65["steps"],
                                           "parity": r65["parity"], "single_stream": r65["single"], "hand_off": r65["hand_off"], "roofline": r65["roof"]}
        except BaseException as e:                   # (SystemExit from a guard included: the headline survives)
            if rank == 0:
                headline["config4_65B"] = {"error": repr(e)}
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(headline), flush=True)
    watchdog.cancel()
    os.dup2(2, 1)                                # communicator teardown may print as well
    dist.destroy_process_group()

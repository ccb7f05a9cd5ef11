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


def make_groups(dist, world: int):
    """The process groups of the decode schedules (module docstring), for ANY backend -- the `nccl` (RCCL) branch of the bench and the
    `gloo` branch of the CPU tests call this one function, so they differ in the transport only:
      token     one communicator over all ranks for the feedback edge (last stage -> stage 0), apart from the forward edges
      fwd[p]    the forward edge r -> r + 1 travels on fwd[r % 2]: a rank's receive (sender r - 1) and its send (sender r) never share
                a communicator
    Returns (token_group, [fwd_even, fwd_odd], topology) where `topology` is a plain description (ranks per group, the parity rule as a
    table edge -> group index) that tests compare across backends."""
    ranks = list(range(world))
    token_group = dist.new_group(ranks)
    fwd_groups = [dist.new_group(ranks), dist.new_group(ranks)]
    topology = {"world": world, "token": {"ranks": ranks, "edge": [world - 1, 0]},
                "fwd": [{"ranks": ranks, "edges": [[r, r + 1] for r in range(world - 1) if r % 2 == p]} for p in (0, 1)]}
    return token_group, fwd_groups, topology


def fwd_group_of(sender: int) -> int:
    """Index into make_groups' fwd list for the edge sender -> sender + 1 (the parity rule, in one place)."""
    return sender % 2


def transport_selfcheck(dist, torch, rank: int, world: int, device: str, backend: str, token_group, fwd_groups, log=None):
    """What the first multi-GPU run should not be the first to execute: one barrier, one object all-gather, and one tensor all-reduce on
    EACH of the three communicators (creates them: RCCL builds a communicator at its first collective).  Returns the gathered
    (rank, device, backend) records; raises what the transport raises."""
    dist.barrier()
    recs = [None] * world
    dist.all_gather_object(recs, {"rank": rank, "device": device, "backend": backend})
    for name, g in (("token", token_group), ("fwd_even", fwd_groups[0]), ("fwd_odd", fwd_groups[1])):
        t = torch.full((1,), rank + 1, dtype=torch.int32, device=device)
        dist.all_reduce(t, group=g)
        if device != "cpu":
            torch.cuda.synchronize()
        want = world * (world + 1) // 2
        if int(t.item()) != want:
            raise RuntimeError(f"transport self-check: all_reduce on group {name} gave {int(t.item())}, expected {want}")
        if log:
            log(f"[pipeline] rank {rank}/{world}: backend {backend}: communicator '{name}' up (all_reduce ok)")
    if log:
        log(f"[pipeline] rank {rank}/{world}: backend {backend}: barrier + all_gather_object ok: {recs}")
    return recs


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
        try:
            for s in range(n_seq):
                _, _, hh, th = self.model.stage_mailbox(s)
                mine.append((hh, th))
        except Exception as e:                  # (allocation / IPC export refused on this rank only: still take part in the collective)
            mine = None
            local_err = e
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if any(x is None for x in everyone):    # ... and every rank draws the same conclusion
            raise RuntimeError("mailboxes unavailable on rank(s) " + ", ".join(str(r) for r, x in enumerate(everyone) if x is None) +
                               (f" ({type(local_err).__name__}: {local_err})" if mine is None else ""))
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
        if not hasattr(self, "tok_all"):
            # one tensor per kind, row s = slot s: consecutive slots form a contiguous block, so a set of sequences is ONE send / receive
            S, dev = self.model.n_seq, self.device
            self.tok_all = torch.zeros(S, dtype=torch.int32, device=dev)
            # a whole-model stage feeds its own pick back: same buffer
            self.tok_out_all = self.tok_all if (self.is_first and self.is_last) else torch.zeros(S, dtype=torch.int32, device=dev)
            self.hid_in_all = None if self.is_first else torch.zeros(S, self.n_embd, dtype=torch.float32, device=dev)
            self.hid_out_all = None if self.is_last else torch.zeros(S, self.n_embd, dtype=torch.float32, device=dev)
            self.tok_in = [self.tok_all[s:s + 1] for s in range(S)]
            self.tok_out = [self.tok_out_all[s:s + 1] for s in range(S)]
            self.hid_in = [None if self.is_first else self.hid_in_all[s] for s in range(S)]
            self.hid_out = [None if self.is_last else self.hid_out_all[s] for s in range(S)]
        self.tok_in[seq].fill_(int(first_token))
        torch.cuda.current_stream().synchronize()
        ptr = lambda t: 0 if t is None else t.data_ptr()
        self.model.stage_bind(seq, n_past,
                              token_in=ptr(self.tok_in[seq]) if self.is_first else 0,
                              hidden_in=ptr(self.hid_in[seq]), hidden_out=ptr(self.hid_out[seq]),
                              token_out=ptr(self.tok_out[seq]) if self.is_last else 0)

    def step(self, seq):
        self.model.stage_step(seq, self.n_threads, self._torch.cuda.current_stream().cuda_stream)

    def step_set(self, seqs):
        """One decode step for all of `seqs` at once (llamahip_stage_step_set: weights streamed once for the set); handles / thread
        counts the set step does not cover (n_threads > 32, head sizes off the grid) step the slots one by one -- same tokens and KV rows;
        stage_logits then holds only the LAST slot's logits, in row 0 (include/llamahip.h)."""
        seqs = list(seqs)
        if not self.model.stage_set_applies(len(seqs), self.n_threads):
            for s in seqs:
                self.step(s)
            return
        self.model.stage_step_set(seqs, self.n_threads, self._torch.cuda.current_stream().cuda_stream)

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
                    fwd_groups=None, token_group=None, host_sync=None):
    """`rounds` greedy tokens for each of `n_seq` bound sequences (stage.bind), one token per sequence
    per round, with no host synchronisation: every receive, stage step and send is enqueued in program
    order and ordered on the device.  Round 0 evaluates the token already in ``stage.tok_in`` (from
    bind or from the previous call); the call ends with stage 0 receiving (stream-ordered) the last
    round's picks into ``tok_in``, so calls can be chained and no send is left unmatched.
    Returns nothing: read the picks with ``stage.trace`` on the last stage."""
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    grp = (lambda sender: fwd_groups[fwd_group_of(sender)]) if fwd_groups else (lambda sender: None)
    sent = [None] * n_seq                         # a sequence's send is waited for before its next step rewrites the buffer (host_sync: see pipeline_decode_sets)

    def recv(t, src, group):
        dist.recv(t, src=src, group=group)
        if host_sync:
            host_sync()

    for k in range(rounds):
        for s in range(n_seq):
            if stage.is_first:
                if world > 1 and k > 0:
                    recv(stage.tok_in[s], world - 1, token_group)     # pick of the previous round
            else:
                recv(stage.hid_in[s], prv, grp(prv))
            if sent[s] is not None:
                sent[s].wait()
                sent[s] = None
            stage.step(s)
            if host_sync and world > 1:
                host_sync()
            if not stage.is_last:
                sent[s] = dist.isend(stage.hid_out[s], dst=nxt, group=grp(rank))
            elif world > 1:
                sent[s] = dist.isend(stage.tok_out[s], dst=0, group=token_group)
    if stage.is_first and world > 1 and rounds > 0:
        for s in range(n_seq):
            recv(stage.tok_in[s], world - 1, token_group)
    for w in sent:
        if w is not None:
            w.wait()


def pipeline_decode_sets(stage: Stage, rank: int, world: int, dist, groups: Sequence[Sequence[int]], rounds: int,
                         fwd_groups=None, token_group=None, host_sync=None):
    """The micro-batched schedule: the bound sequences are split into `groups` (consecutive slots each); a stage takes
    a whole group per step -- ONE receive of the group's rows, ONE llamahip_stage_step_set (the stage's weights are streamed once for
    all of its sequences), ONE send -- so with as many groups as stages every stage is busy and every weight byte serves a group's
    worth of tokens.  Same contract as pipeline_decode: round 0 evaluates the tokens bound into ``stage.tok_in``, the call ends with
    stage 0 having received the last round's picks; results through ``stage.trace``.
    A group's send is waited for (stream-side with RCCL) before the group's next step rewrites the buffer it reads.  `host_sync`: a
    transport that is not ordered on the device stream (gloo moving CUDA tensors: the one-GPU smoke test) gets a host
    synchronisation before every send and after every receive."""
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    grp = (lambda sender: fwd_groups[fwd_group_of(sender)]) if fwd_groups else (lambda sender: None)
    sent = [None] * len(groups)
    for seqs in groups:
        # a group's rows travel as ONE slice [seqs[0], seqs[-1] + 1) of the per-slot buffers: a group must be a run of consecutive slots in
        # ascending order (step_set itself takes any order; a reordered group would move the wrong rows between stages, silently)
        if len(seqs) == 0 or list(seqs) != list(range(seqs[0], seqs[0] + len(seqs))):
            raise ValueError(f"pipeline_decode_sets: group {list(seqs)} is not a run of consecutive slots in ascending order")

    def recv(t, src, group):
        dist.recv(t, src=src, group=group)
        if host_sync:
            host_sync()

    for k in range(rounds):
        for gi, seqs in enumerate(groups):
            lo, hi = seqs[0], seqs[-1] + 1
            if stage.is_first:
                if world > 1 and k > 0:
                    recv(stage.tok_all[lo:hi], world - 1, token_group)      # the group's picks of the previous round
            else:
                recv(stage.hid_in_all[lo:hi], prv, grp(prv))
            if sent[gi] is not None:
                sent[gi].wait()
                sent[gi] = None
            if len(seqs) > 1:
                stage.step_set(seqs)
            else:
                stage.step(seqs[0])
            if host_sync and world > 1:
                host_sync()
            if not stage.is_last:
                sent[gi] = dist.isend(stage.hid_out_all[lo:hi], dst=nxt, group=grp(rank))
            elif world > 1:
                sent[gi] = dist.isend(stage.tok_out_all[lo:hi], dst=0, group=token_group)
    if stage.is_first and world > 1 and rounds > 0:
        for seqs in groups:
            recv(stage.tok_all[seqs[0]:seqs[-1] + 1], world - 1, token_group)
    for w in sent:
        if w is not None:
            w.wait()


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

"""The N > 1 path of bench.py: the layer-pipeline schedule (llama.swift_amd/pipeline.py).

CPU (world_size 2 and 3, gloo): the schedule itself -- ordering, non-blocking hand-off, token
feedback, per-sequence KV slots -- with oracle-backed stages standing in for the GPUs; the pipelined
tokens must equal the monolithic greedy decode of every sequence.
GPU (-m gpu, single device): the product stage (llamahip_eval_stage + KV sequence slots) split in two
on one GPU must reproduce the whole-model logits bit for bit."""
import os
import sys

import numpy as np
import pytest

import synth


def same(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_layer_range_partitions_every_layer_once(L):
    from llama_swift_amd.pipeline import layer_range
    for n_layer, world in [(32, 1), (32, 8), (80, 8), (40, 3), (5, 4), (60, 7)]:
        got = [layer_range(n_layer, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == n_layer
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
        assert max(hi - lo for lo, hi in got) - min(hi - lo for lo, hi in got) <= 1


class OracleStage:
    """CPU stand-in for a GPU stage: layers [lo, hi) evaluated by the oracle, one model instance per
    in-flight sequence (= one KV cache per sequence)."""

    def __init__(self, path, n_ctx, rank, world, n_seq, n_layer):
        import reflib
        from llama_swift_amd.pipeline import layer_range
        self.lo, self.hi = layer_range(n_layer, rank, world)
        lib = reflib.OracleLib()
        self.models = [lib.load(path, n_ctx) for _ in range(n_seq)]
        self.is_first, self.is_last = self.lo == 0, self.hi == n_layer
        self.n_embd, self.n_vocab = self.models[0].n_embd, self.models[0].n_vocab
        self.device = "cpu"

    def run(self, seq, n_past, tokens, hidden):
        import torch
        hin = None if self.is_first else hidden.numpy()
        hout, logits = self.models[seq].eval_range(self.lo, self.hi, n_past, tokens=tokens if self.is_first else None, hidden_in=hin, n_threads=8)
        return logits if self.is_last else torch.from_numpy(hout)

    # single-token steps on bound sequences (pipeline_decode); synchronous on the CPU
    def bind(self, seq, n_past, first_token):
        import torch
        if not hasattr(self, "tok_in"):
            S = len(self.models)
            self.tok_all = torch.zeros(S, dtype=torch.int32)
            self.tok_out_all = self.tok_all if (self.is_first and self.is_last) else torch.zeros(S, dtype=torch.int32)
            self.hid_in_all = None if self.is_first else torch.zeros(S, self.n_embd)
            self.hid_out_all = None if self.is_last else torch.zeros(S, self.n_embd)
            self.tok_in = [self.tok_all[s:s + 1] for s in range(S)]
            self.tok_out = [self.tok_out_all[s:s + 1] for s in range(S)]
            self.hid_in = [None if self.is_first else self.hid_in_all[s] for s in range(S)]
            self.hid_out = [None if self.is_last else self.hid_out_all[s] for s in range(S)]
            self.pos, self.picked = [0] * S, [[] for _ in range(S)]
        self.tok_in[seq].fill_(int(first_token))
        self.pos[seq], self.picked[seq] = n_past, []

    def step(self, seq):
        import torch
        tk = self.tok_in[seq].numpy().copy() if self.is_first else None
        hin = None if self.is_first else self.hid_in[seq].numpy()
        hout, logits = self.models[seq].eval_range(self.lo, self.hi, self.pos[seq], tokens=tk, hidden_in=hin, n_threads=8)
        self.pos[seq] += 1
        if self.is_last:
            t = int(np.argmax(logits))
            self.picked[seq].append(t)
            self.tok_out[seq].fill_(t)
        else:
            self.hid_out[seq].copy_(torch.from_numpy(hout).reshape(-1))

    def step_set(self, seqs):
        for s in seqs:                              # (the CPU stand-in has nothing to batch: one eval per sequence is the definition)
            self.step(s)

    def trace(self, seq, cap):
        return len(self.picked[seq]), self.pos[seq], np.array(self.picked[seq][:cap], np.int32)


def _worker(rank, world, path, n_layer, init_file, rounds, out_file):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from llama_swift_amd.pipeline import gather_traces, pipeline_decode, pipeline_rounds
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    token_group = dist.new_group(list(range(world)))
    fwd_groups = [dist.new_group(list(range(world))), dist.new_group(list(range(world)))]
    S = world + 1                                  # more sequences than stages also has to work
    stage = OracleStage(path, 64, rank, world, S, n_layer)
    prompts = [synth.synth_prompt(5 + s, 96, seed=10 + s) for s in range(S)]
    toks, n_past = pipeline_rounds(stage, rank, world, dist, torch, prompts, [0] * S, rounds, token_group)
    toks2, n_past = pipeline_rounds(stage, rank, world, dist, torch, [np.array([toks[s, -1]], np.int32) for s in range(S)], n_past, 3, token_group)
    # ... and the stream-ordered decode schedule continues every sequence: 2 + 3 chained rounds
    for s in range(S):
        stage.bind(s, n_past[s], int(toks2[s, -1]))
    pipeline_decode(stage, rank, world, dist, S, 2, fwd_groups, token_group)
    pipeline_decode(stage, rank, world, dist, S, 3, fwd_groups, token_group)
    # ... and the micro-batched schedule (sets of consecutive slots, one message per set): 2 more rounds
    from llama_swift_amd.pipeline import pipeline_decode_sets
    groups = [list(range(0, 2)), list(range(2, S))]
    pipeline_decode_sets(stage, rank, world, dist, groups, 2, fwd_groups, token_group)
    toks3, pos = gather_traces(stage, rank, world, dist, torch, S, 7)
    assert pos == [n + 7 for n in n_past], (rank, pos, n_past)
    if rank == 0:
        np.savez(out_file, toks=np.concatenate([toks, toks2, toks3], axis=1), n_past=np.array(pos))
    dist.barrier()
    dist.destroy_process_group()


def _failing_worker(rank, world, path, n_layer, init_file, mode, out_dir):
    """rank 1 goes away (mode "exit": the process dies; mode "hang": it stops taking part) after the prompt round."""
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import time

    import torch
    import torch.distributed as dist
    from llama_swift_amd.pipeline import PipelineError, pipeline_rounds, run_guarded
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    token_group = dist.new_group(list(range(world)))
    stage = OracleStage(path, 64, rank, world, 2, n_layer)
    prompts = [synth.synth_prompt(5 + s, 96, seed=10 + s) for s in range(2)]
    toks, n_past = pipeline_rounds(stage, rank, world, dist, torch, prompts, [0, 0], 1, token_group)
    if rank == 1:
        if mode == "exit":
            os._exit(7)
        time.sleep(3600)                           # hang: alive, silent
    last = [np.array([toks[s, -1]], np.int32) for s in range(2)]
    try:
        run_guarded(lambda: pipeline_rounds(stage, rank, world, dist, torch, last, n_past, 4, token_group), rank, world, 6.0, "pipeline_rounds (decode)")
    except PipelineError as e:
        open(os.path.join(out_dir, "rank0.err"), "w").write(f"{e.code} {e.message}")
        os._exit(4)
    os._exit(0)                                    # must not happen: the peer is gone


@pytest.mark.parametrize("mode", ["exit", "hang"])
def test_pipeline_peer_failure_is_reported_not_hung(built, tmp_path, mode):
    """A stage whose peer dies or hangs must end with a non-zero exit code and a PredictionFailed (-1001) message within
    the watchdog limit -- never wait forever (HIP/RCCL errors map to -1001 during eval; the bridge posts
    `failed` once)."""
    import multiprocessing as mp
    import time
    hp = synth.HParams(n_vocab=96, n_embd=256, n_mult=64, n_head=2, n_layer=2)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=31))
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, path, 2, str(tmp_path / "rdv"), mode, str(tmp_path))) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    procs[0].join(90)
    took = time.time() - t0
    alive = procs[0].is_alive()
    for p in procs:
        if p.is_alive():
            p.kill()
        p.join()
    assert not alive, "rank 0 was still waiting for its dead peer after 90 s"
    assert procs[0].exitcode in (3, 4), procs[0].exitcode           # 4: transport error -> PipelineError; 3: watchdog
    if procs[0].exitcode == 4:
        msg = open(str(tmp_path / "rank0.err")).read()
        assert msg.startswith("-1001 ") and "rank 0/2" in msg, msg
    assert took < 80


def _mailbox_wiring_worker(rank, world, init_file, n_seq, out_dir):
    """HipStage.setup_mailboxes with a recording stand-in for the model handle: which inbox handle does each rank open?"""
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import json

    import torch.distributed as dist
    from llama_swift_amd.pipeline import HipStage
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)

    class FakeModel:
        def __init__(self):
            self.connected = {}

        def stage_mailbox(self, s):          # (hidden inbox ptr, token inbox ptr, hidden handle, token handle): 64-byte handles naming (rank, slot, kind)
            tag = lambda kind: (b"%c%03d%03d" % (kind, rank, s)).ljust(64, b".")
            return 0, 0, (tag(ord("H")) if rank > 0 else None), (tag(ord("T")) if rank == 0 and world > 1 else None)

        def stage_mailbox_connect(self, s, next_hidden_handle=None, next_hidden_ptr=0, token_handle=None, token_ptr=0):
            self.connected[s] = (None if next_hidden_handle is None else bytes(next_hidden_handle)[:7].decode(), None if token_handle is None else bytes(token_handle)[:7].decode())

    class Stub:
        model = FakeModel()
    st = Stub()
    HipStage.setup_mailboxes(st, dist, rank, world, n_seq)
    assert st.mailboxes is True
    json.dump({str(k): v for k, v in st.model.connected.items()}, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_mailbox_handle_exchange_wires_every_stage_to_its_successor_gloo(tmp_path, world):
    """The one collective of the device-side hand-off (HipStage.setup_mailboxes: an object all-gather of the 64-byte IPC handles): for
    every sequence slot, stage r opens the hidden inbox of stage r + 1 and nothing else, the last stage also the first stage's token
    inbox -- for worlds larger than the two ranks the one-GPU tests can run."""
    import json

    import torch.multiprocessing as mp
    n_seq = 3
    mp.spawn(_mailbox_wiring_worker, args=(world, str(tmp_path / "rdv"), n_seq, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = json.load(open(tmp_path / f"rank{r}.json"))
        for s in range(n_seq):
            hid, tok = got[str(s)]
            assert hid == (f"H{r + 1:03d}{s:03d}" if r + 1 < world else None), (world, r, s, hid)
            assert tok == (f"T000{s:03d}" if r == world - 1 else None), (world, r, s, tok)


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_schedule_gloo(built, tmp_path, world):
    import torch.multiprocessing as mp
    n_layer = 3
    hp = synth.HParams(n_vocab=96, n_embd=256, n_mult=64, n_head=2, n_layer=n_layer)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=31))
    rounds = 4
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, path, n_layer, str(tmp_path / "rdv"), rounds, out), nprocs=world, join=True)
    got = np.load(out)
    import reflib
    lib = reflib.OracleLib()
    S = world + 1
    for s in range(S):
        m = lib.load(path, 64)
        prompt = synth.synth_prompt(5 + s, 96, seed=10 + s)
        lg = m.eval(prompt, 0, 8)["logits"]
        n_past, want = len(prompt), []
        for _ in range(rounds + 3 + 7):
            t = int(np.argmax(lg)); want.append(t)
            lg = m.eval(np.array([t], np.int32), n_past, 8)["logits"]; n_past += 1
        assert got["toks"][s].tolist() == want, f"sequence {s}"
        assert int(got["n_past"][s]) == len(prompt) + rounds - 1 + 3 + 7



# ---- the N > 1 bench leg's control flow at world 8 (bench.pipeline_bench_main) on the CPU: gloo + oracle stages ----
class _CpuEnv:
    """bench.CudaEnv's shape for the CPU: no streams, oracle stages (one llama_eval per sequence and token)."""
    device = "cpu"

    def sync(self):
        pass

    def lane(self):
        return None

    def on(self, lane):
        import contextlib
        return contextlib.nullcontext()

    def make_stage(self, path, n_ctx, rank, world, n_seq, n_layer, n_threads):
        return OracleStage(path, n_ctx, rank, world, n_seq, n_layer)

    def stage_roofline(self, *a, **k):
        return {"note": "CPU control-flow test: no roofline"}

    def close_stage(self, stage):
        pass


def _bench_main_worker(rank, world, init_file, out_dir, env_extra):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import argparse
    import json

    import torch.distributed as dist
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "LLAMAHIP_PIPE_WATCHDOG_S": "600", "LLAMAHIP_PIPE_PARITY_S": "20"})
    os.environ.update(env_extra)
    import bench
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    cfgs = {"tiny8": dict(n_vocab=96, n_embd=64, n_mult=32, n_head=1, n_layer=8), "65B": dict(n_vocab=96, n_embd=64, n_mult=32, n_head=1, n_layer=80)}

    def model_path(name, cfg, seed):
        path = os.path.join(out_dir, f"{name}.bin")
        if not os.path.exists(path):
            hp = synth.HParams(**cfg)
            synth.write_model(path + ".tmp", hp, synth.random_tensors(hp, seed=seed % 1000))
            os.replace(path + ".tmp", path)
        return path
    args = argparse.Namespace(gpus=world, steps=3, warmup=1, model="tiny8", n_ctx=64, threads=8, seed=20230312)
    def emit(line):                                 # (written at once: a failing extra leg ends the process right after emitting)
        json.dump([line], open(os.path.join(out_dir, "line.json"), "w"))
    bench.pipeline_bench_main(args, cfgs["tiny8"], model_path, lambda *a: None, cfgs, env=_CpuEnv(), emit=emit)
    dist.barrier()
    dist.destroy_process_group()


def _expected_topology(world):
    """The communicators of the decode schedules, written down independently of pipeline.make_groups: one over all ranks for the token
    feedback edge (last -> 0), two over all ranks for the forward edges r -> r + 1 by parity of the SENDER."""
    ranks = list(range(world))
    return {"world": world, "token": {"ranks": ranks, "edge": [world - 1, 0]},
            "fwd": [{"ranks": ranks, "edges": [[r, r + 1] for r in range(0, world - 1, 2)]}, {"ranks": ranks, "edges": [[r, r + 1] for r in range(1, world - 1, 2)]}]}


def test_nccl_and_gloo_branches_build_the_same_process_groups(built):
    """bench.pipeline_bench_main creates its communicators through pipeline.make_groups whatever the backend, so the RCCL branch (never
    run on more than one GPU so far) differs from the gloo branch the CPU tests run in the transport only: the function issues exactly
    three new_group calls over all ranks in a fixed order on ANY `dist`, its topology is the independent specification above, the
    schedules pick a forward group through the same parity rule, and the bench leg has no new_group call of its own."""
    import inspect

    sys.path.insert(0, ROOT)
    import bench
    from llama_swift_amd import pipeline

    class RecordingDist:
        def __init__(self):
            self.calls = []

        def new_group(self, ranks):
            self.calls.append(list(ranks))
            return ("group", len(self.calls) - 1)

    for world in (1, 2, 3, 8):
        d = RecordingDist()
        tok, fwd, topo = pipeline.make_groups(d, world)
        assert d.calls == [list(range(world))] * 3 and tok == ("group", 0) and fwd == [("group", 1), ("group", 2)]
        assert topo == _expected_topology(world)
        for r in range(world - 1):
            assert [r, r + 1] in topo["fwd"][pipeline.fwd_group_of(r)]["edges"]
            # a rank's receive (sender r - 1) and its send (sender r) never share a communicator
            assert r == 0 or pipeline.fwd_group_of(r - 1) != pipeline.fwd_group_of(r)
    src = inspect.getsource(bench.pipeline_bench_main)
    assert "make_groups(dist, world)" in src and "new_group(" not in src and "transport_selfcheck(" in src
    for fn in (pipeline.pipeline_decode, pipeline.pipeline_decode_sets):
        assert "fwd_group_of(sender)" in inspect.getsource(fn)


@pytest.mark.parametrize("mode", ["sets", "one_per_step"])
def test_bench_pipeline_control_flow_at_world_8_gloo(built, tmp_path, mode):
    """No 8-GPU node has ever run this code, so its control flow runs here: `bench.py --gpus 8` (bench.pipeline_bench_main) with eight
    gloo ranks and oracle stages -- prompt rounds, (one_per_step: mailbox setup that fails on every rank -> the common fall-back to the
    per-token hand-off; sets: groups of sequences per stage, one message per set), bind + barrier, warm-up, timed loop, the
    single-stream leg, trace gather, the parity gate against the CPU path, then the 65B leg's layer_range(80, r, 8) partition with
    its own parity gate -- and ONE line comes out on rank 0 with both legs parity-identical."""
    import json

    import torch.multiprocessing as mp
    world = 8
    env_extra = {"LLAMAHIP_PIPE_SEQS_PER_STAGE": "2", "LLAMAHIP_PIPE_SET": "1" if mode == "sets" else "0", "LLAMAHIP_PIPE_65B_STEPS": "2", "ORC_OMP_THREADS": "1"}
    mp.spawn(_bench_main_worker, args=(world, str(tmp_path / "rdv"), str(tmp_path), env_extra), nprocs=world, join=True)
    lines = json.load(open(tmp_path / "line.json"))
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 8 and line["config"]["sequences"] == 16 and line["value"] > 0
    assert line["parity"]["checked"] and line["parity"]["identical"], line["parity"]
    assert ("one set" in line["config"]["workload"]) == (mode == "sets")
    assert ("one message per set" in line["config"]["hand_off"]) == (mode == "sets")
    # the line names the transport that actually ran, and the communicators are the ones every backend gets (next test)
    assert line["config"]["backend"] == "gloo" and "gloo" in line["config"]["hand_off"] and "RCCL" not in line["config"]["hand_off"]
    assert line["config"]["group_topology"] == _expected_topology(8) and "ok on 8 rank(s)" in line["config"]["transport_selfcheck"]
    leg = line["config4_65B"]
    assert "error" not in leg, leg
    assert "80 layers over 8 stages" in leg["workload"] and leg["parity"]["checked"] and leg["parity"]["identical"], leg
    assert line["single_stream"]["tokens"] == 16

@pytest.mark.gpu
def test_hip_stage_in_the_pipeline_schedule_single_rank(L, tmp_path):
    """world_size 1 on the real device: HipStage (llamahip_eval_stage, KV sequence slots) driven by
    the same schedule the multi-GPU bench uses, against the device-resident greedy loop."""
    import torch

    from llama_swift_amd.pipeline import HipStage, pipeline_rounds
    hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=3)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=17))
    S = 3
    stage = HipStage(path, 64, 0, 1, S, hp.n_layer, 0)
    prompts = [synth.synth_prompt(4 + s, hp.n_vocab, seed=40 + s) for s in range(S)]
    toks, n_past = pipeline_rounds(stage, 0, 1, None, torch, prompts, [0] * S, 6)
    assert n_past == [len(p) + 5 for p in prompts]
    with L.Model(path, n_ctx=64) as whole:
        for s in range(S):
            lg = whole.eval(prompts[s], 0, 8)
            first = int(np.argmax(lg))
            rest = whole.decode_greedy(first, len(prompts[s]), 5, 8)
            assert toks[s].tolist() == [first] + rest.tolist(), f"sequence {s}"
    stage.model.close()


@pytest.mark.gpu
def test_two_stages_on_one_gpu_equal_the_whole_model(L, tmp_path):
    import torch
    hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=4)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=9))
    whole = L.Model(path, n_ctx=64)
    a = L.Model(path, n_ctx=64, layer_begin=0, layer_end=1, n_seq=2)        # uneven split on purpose
    b = L.Model(path, n_ctx=64, layer_begin=1, layer_end=4, n_seq=2)
    d = hp.n_embd
    prompts = [synth.synth_prompt(9, hp.n_vocab, seed=1), synth.synth_prompt(6, hp.n_vocab, seed=2)]
    n_past = [0, 0]
    for step in range(5):
        for s in (1, 0):                                                     # interleave the two sequences
            toks = prompts[s] if step == 0 else np.array([nxt[s]], np.int32)
            if step == 0 and s == 1:
                nxt = [0, 0]
            h = torch.empty(len(toks) * d, dtype=torch.float32, device="cuda")
            a.set_seq(s); b.set_seq(s)
            a.eval_stage(n_past[s], tokens=toks, hidden_out=h.data_ptr())
            lg = b.eval_stage(n_past[s], n_tokens=len(toks), hidden_in=h.data_ptr(), want_logits=True)
            n_past[s] += len(toks)
            nxt[s] = int(np.argmax(lg))
            if s == 0:                                                       # the whole model follows sequence 0 only
                want = whole.eval(toks, n_past[0] - len(toks), 8)
                assert np.array_equal(lg.view(np.uint32), want.view(np.uint32)), f"step {step}"
    with pytest.raises(L.LlamaHipError):
        a.set_seq(2)
    with pytest.raises(L.LlamaHipError):
        b.eval([1], 0)                                                        # a stage handle is not a whole model
    for m in (whole, a, b):
        m.close()


@pytest.mark.gpu
def test_stage_handles_vs_the_oracle_layer_ranges(L, oracle, tmp_path):
    """A stage handle against the CPU restatement of the SAME layer range (orc_eval_range), not against the
    whole-model HIP path: the residual stream a stage hands on (hidden_out) and the last stage's logits, bit for
    bit, for a 9-token prompt chunk and for single-token steps (the fused decode schedule), with an uneven split."""
    import torch
    hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=4)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=23))
    om = oracle.load(path, 64)
    cuts = [0, 1, 3, 4]
    stages = [L.Model(path, n_ctx=64, layer_begin=a, layer_end=b) for a, b in zip(cuts[:-1], cuts[1:])]
    d = hp.n_embd
    toks, n_past = synth.synth_prompt(9, hp.n_vocab, seed=6), 0
    for step in range(5):
        N = len(toks)
        h_gpu = [torch.empty(N * d, dtype=torch.float32, device="cuda") for _ in range(2)]
        # CPU: stage by stage
        want_h, hin = [], None
        for si, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            hout, logits = om.eval_range(a, b, n_past, tokens=toks if si == 0 else None, hidden_in=hin, n_threads=8)
            want_h.append(hout); hin = hout
        # GPU: the same chain through llamahip_eval_stage
        stages[0].eval_stage(n_past, tokens=toks, hidden_out=h_gpu[0].data_ptr())
        stages[1].eval_stage(n_past, n_tokens=N, hidden_in=h_gpu[0].data_ptr(), hidden_out=h_gpu[1].data_ptr())
        lg = stages[2].eval_stage(n_past, n_tokens=N, hidden_in=h_gpu[1].data_ptr(), want_logits=True)
        for si in range(2):
            got = h_gpu[si].cpu().numpy()
            assert np.array_equal(got.view(np.uint32), want_h[si].view(np.uint32)), f"step {step}: hidden_out of stage {si} differs from orc_eval_range"
        assert np.array_equal(lg.view(np.uint32), logits.view(np.uint32)), f"step {step}: logits"
        n_past += N
        toks = np.array([int(np.argmax(logits))], np.int32)
    for m in stages:
        m.close()
    om.close()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 1])          # hipGraph replay / eager launches on the caller's stream
def test_stream_ordered_stage_steps_single_rank(L, tmp_path, flags):
    """pipeline_decode on one rank: llamahip_stage_bind/step/trace (device-side position and greedy
    pick, caller's stream) must continue a sequence exactly like the device-resident greedy loop."""
    import torch

    from llama_swift_amd.pipeline import HipStage, gather_traces, pipeline_decode, pipeline_rounds
    hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=3)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=17))
    S = 3
    stage = HipStage(path, 64, 0, 1, S, hp.n_layer, 0)
    stage.model.close()
    stage.model = L.Model(path, n_ctx=64, device=0, n_seq=S, flags=flags)
    prompts = [synth.synth_prompt(4 + s, hp.n_vocab, seed=40 + s) for s in range(S)]
    toks, n_past = pipeline_rounds(stage, 0, 1, None, torch, prompts, [0] * S, 1)
    for s in range(S):
        stage.bind(s, n_past[s], int(toks[s, -1]))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                  # not the default stream: the step must follow the caller's
        pipeline_decode(stage, 0, 1, None, S, 4)
        pipeline_decode(stage, 0, 1, None, S, 3)
    got, pos = gather_traces(stage, 0, 1, None, torch, S, 7)
    assert pos == [len(p) + 7 for p in prompts]
    with L.Model(path, n_ctx=64) as whole:
        for s in range(S):
            first = int(np.argmax(whole.eval(prompts[s], 0, 8)))
            assert first == int(toks[s, -1])
            rest = whole.decode_greedy(first, len(prompts[s]), 7, 8)
            assert got[s].tolist() == rest.tolist(), f"sequence {s}"
    # stepping past the context is refused on the host
    stage.bind(0, 63, 1)
    stage.step(0)
    with pytest.raises(L.LlamaHipError, match="context overflow"):
        stage.step(0)
    with pytest.raises(L.LlamaHipError, match="not bound"):
        stage.model.stage_step(S + 5)
    stage.model.close()


@pytest.mark.gpu
def test_stream_ordered_two_stages_on_one_gpu(L, tmp_path):
    """Two stage handles chained on one stream through shared device buffers (what the RCCL
    send/recv pair does between GPUs): embed+layer 0 -> layers 1..3 + lm head + pick -> token slot."""
    import torch
    hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=4)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=9))
    a = L.Model(path, n_ctx=64, layer_begin=0, layer_end=1, n_seq=2)
    b = L.Model(path, n_ctx=64, layer_begin=1, layer_end=4, n_seq=2)
    whole = L.Model(path, n_ctx=64)
    prompts = [synth.synth_prompt(9, hp.n_vocab, seed=1), synth.synth_prompt(6, hp.n_vocab, seed=2)]
    tok = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(2)]
    hid = [torch.zeros(hp.n_embd, dtype=torch.float32, device="cuda") for _ in range(2)]
    firsts = []
    for s in range(2):                              # prompts through the host-synchronous entry point
        h = torch.empty(len(prompts[s]) * hp.n_embd, dtype=torch.float32, device="cuda")
        a.set_seq(s); b.set_seq(s)
        a.eval_stage(0, tokens=prompts[s], hidden_out=h.data_ptr())
        lg = b.eval_stage(0, n_tokens=len(prompts[s]), hidden_in=h.data_ptr(), want_logits=True)
        firsts.append(int(np.argmax(lg)))
        tok[s].fill_(firsts[s])
        torch.cuda.synchronize()
        a.stage_bind(s, len(prompts[s]), token_in=tok[s].data_ptr(), hidden_out=hid[s].data_ptr())
        b.stage_bind(s, len(prompts[s]), hidden_in=hid[s].data_ptr(), token_out=tok[s].data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(6):
        for s in (1, 0):
            a.stage_step(s, 8, st)
            b.stage_step(s, 8, st)
    for s in range(2):
        n, pos, got = b.stage_trace(s, 6)
        assert (n, pos) == (6, len(prompts[s]) + 6)
        assert a.stage_trace(s, 0)[:2] == (6, len(prompts[s]) + 6)
        assert int(np.argmax(whole.eval(prompts[s], 0, 8))) == firsts[s]
        want = whole.decode_greedy(firsts[s], len(prompts[s]), 6, 8)
        assert got.tolist() == want.tolist(), f"sequence {s}"
    with pytest.raises(L.LlamaHipError, match="hidden_in"):
        b.stage_bind(0, 0, token_in=tok[0].data_ptr())
    for m in (whole, a, b):
        m.close()


# ------------------------------------------------------------------------------------------------ device-side mailboxes
def _mailbox_model(tmp_path, shape):
    """small: the separate wq|wk|wv + attention launches (k_gemv PREP_NORM_TAG); 7b_width: k_qkv_attn's tagged prologue."""
    from conftest import synth_tool
    if shape == "small":
        hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=4)
        path = str(tmp_path / "m.bin")
        synth.write_model(path, hp, synth.random_tensors(hp, seed=9))
        return path, hp.n_vocab, hp.n_embd, 4
    path = synth_tool(tmp_path / "w.bin", seed=12, n_vocab=512, n_embd=4096, n_mult=256, n_head=32, n_layer=3)
    return path, 512, 4096, 3


def _prompt_through_stages(torch, a, b, prompts, n_embd):
    firsts = []
    for s, p in enumerate(prompts):
        h = torch.empty(len(p) * n_embd, dtype=torch.float32, device="cuda")
        a.set_seq(s); b.set_seq(s)
        a.eval_stage(0, tokens=p, hidden_out=h.data_ptr())
        lg = b.eval_stage(0, n_tokens=len(p), hidden_in=h.data_ptr(), want_logits=True)
        firsts.append(int(np.argmax(lg)))
    torch.cuda.synchronize()
    return firsts


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["small", "7b_width"])
def test_device_side_mailboxes_two_stages_two_streams(L, tmp_path, shape):
    """Two stage handles in one process hand the residual-stream row and the token over through device-side mailboxes (tagged
    granules stored by the last kernel of a stage step, polled by the first kernel of the neighbour's): no shared hidden / token
    buffers, no ordering between the two streams -- stage B's steps are enqueued BEFORE stage A's, so its kernels really wait.
    Tokens must be those of the whole model's greedy loop."""
    import torch
    path, n_vocab, n_embd, n_layer = _mailbox_model(tmp_path, shape)
    a = L.Model(path, n_ctx=64, layer_begin=0, layer_end=1, n_seq=2)
    b = L.Model(path, n_ctx=64, layer_begin=1, layer_end=n_layer, n_seq=2)
    whole = L.Model(path, n_ctx=64)
    prompts = [synth.synth_prompt(9, n_vocab, seed=1), synth.synth_prompt(6, n_vocab, seed=2)]
    firsts = _prompt_through_stages(torch, a, b, prompts, n_embd)
    tok = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(2)]
    for s in range(2):
        _, a_tok, _, _ = a.stage_mailbox(s)                       # stage A: token inbox
        b_hid, _, _, _ = b.stage_mailbox(s)                       # stage B: hidden inbox
        assert a_tok and b_hid
        a.stage_mailbox_connect(s, next_hidden_ptr=b_hid)
        b.stage_mailbox_connect(s, token_ptr=a_tok)
        tok[s].fill_(firsts[s])
        torch.cuda.synchronize()
        a.stage_bind(s, len(prompts[s]), token_in=tok[s].data_ptr())            # no hidden_out: the mailbox
        b.stage_bind(s, len(prompts[s]))                                          # no hidden_in, no token_out
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    n_steps = 7
    for _ in range(n_steps):
        for s in (1, 0):
            b.stage_step(s, 8, sb.cuda_stream)                   # the consumer first: it polls until A's row arrives
            a.stage_step(s, 8, sa.cuda_stream)
    for s in range(2):
        n, pos, got = b.stage_trace(s, n_steps)
        assert (n, pos) == (n_steps, len(prompts[s]) + n_steps)
        assert a.stage_trace(s, 0)[:2] == (n_steps, len(prompts[s]) + n_steps)
        assert int(np.argmax(whole.eval(prompts[s], 0, 8))) == firsts[s]
        want = whole.decode_greedy(firsts[s], len(prompts[s]), n_steps, 8)
        assert got.tolist() == want.tolist(), f"{shape}: sequence {s}: {got.tolist()} vs {want.tolist()}"
    # re-binding at an earlier position: stale rows of the first run must not be taken for new ones
    for s in range(2):
        tok[s].fill_(firsts[s]); torch.cuda.synchronize()
        a.stage_bind(s, len(prompts[s]), token_in=tok[s].data_ptr())
        b.stage_bind(s, len(prompts[s]))
    for _ in range(3):
        for s in (0, 1):
            b.stage_step(s, 8, sb.cuda_stream)
            a.stage_step(s, 8, sa.cuda_stream)
    for s in range(2):
        n, pos, got = b.stage_trace(s, 3)
        whole.eval(prompts[s], 0, 8)
        assert got.tolist() == whole.decode_greedy(firsts[s], len(prompts[s]), 3, 8).tolist()
    for m in (whole, a, b):
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["small", "7b_width"])
def test_device_side_mailboxes_three_stages_with_a_one_layer_middle_stage(L, tmp_path, shape):
    """What a pipeline of more than two GPUs adds: a MIDDLE stage, whose step takes its row from one mailbox and leaves it in another --
    here with a single layer, so the mailbox prologue (first layer) and the mailbox epilogue (last layer's w2) belong to the same
    layer.  Three stage handles, three streams, the consumers' steps enqueued before their producers'; tokens = the whole model's."""
    import torch
    path, n_vocab, n_embd, n_layer = _mailbox_model(tmp_path, shape)
    a = L.Model(path, n_ctx=64, layer_begin=0, layer_end=1, n_seq=2)
    mid = L.Model(path, n_ctx=64, layer_begin=1, layer_end=2, n_seq=2)
    c = L.Model(path, n_ctx=64, layer_begin=2, layer_end=n_layer, n_seq=2)
    whole = L.Model(path, n_ctx=64)
    prompts = [synth.synth_prompt(9, n_vocab, seed=1), synth.synth_prompt(6, n_vocab, seed=2)]
    firsts = []
    for s, p in enumerate(prompts):
        h1 = torch.empty(len(p) * n_embd, dtype=torch.float32, device="cuda")
        h2 = torch.empty(len(p) * n_embd, dtype=torch.float32, device="cuda")
        for m in (a, mid, c):
            m.set_seq(s)
        a.eval_stage(0, tokens=p, hidden_out=h1.data_ptr())
        mid.eval_stage(0, n_tokens=len(p), hidden_in=h1.data_ptr(), hidden_out=h2.data_ptr())
        firsts.append(int(np.argmax(c.eval_stage(0, n_tokens=len(p), hidden_in=h2.data_ptr(), want_logits=True))))
    torch.cuda.synchronize()
    tok = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(2)]
    for s in range(2):
        _, a_tok, _, _ = a.stage_mailbox(s)
        m_hid, _, _, _ = mid.stage_mailbox(s)
        c_hid, _, _, _ = c.stage_mailbox(s)
        assert a_tok and m_hid and c_hid
        a.stage_mailbox_connect(s, next_hidden_ptr=m_hid)
        mid.stage_mailbox_connect(s, next_hidden_ptr=c_hid)
        c.stage_mailbox_connect(s, token_ptr=a_tok)
        tok[s].fill_(firsts[s])
        torch.cuda.synchronize()
        a.stage_bind(s, len(prompts[s]), token_in=tok[s].data_ptr())
        mid.stage_bind(s, len(prompts[s]))                                        # neither hidden_in nor hidden_out: two mailboxes
        c.stage_bind(s, len(prompts[s]))
    sa, sm, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    n_steps = 6
    for _ in range(n_steps):
        for s in (1, 0):
            if shape == "small":
                c.stage_step(s, 8, sc.cuda_stream)               # consumers first: their kernels really wait
                mid.stage_step(s, 8, sm.cuda_stream)
                a.stage_step(s, 8, sa.cuda_stream)
            else:
                # 7B-wide stages: the waiting first launches of TWO stages (576 polling workgroups each) fill the ONE GPU of this test and the
                # producer's launch would never become resident -- a co-location artefact, every stage has its own GPU in the pipeline.
                # Producer first, one stage at a time: the rows still travel through the mailboxes.
                for m, st in ((a, sa), (mid, sm), (c, sc)):
                    m.stage_step(s, 8, st.cuda_stream)
                    torch.cuda.synchronize()
    for s in range(2):
        n, pos, got = c.stage_trace(s, n_steps)
        assert (n, pos) == (n_steps, len(prompts[s]) + n_steps)
        assert mid.stage_trace(s, 0)[:2] == (n_steps, len(prompts[s]) + n_steps)
        assert int(np.argmax(whole.eval(prompts[s], 0, 8))) == firsts[s]
        want = whole.decode_greedy(firsts[s], len(prompts[s]), n_steps, 8)
        assert got.tolist() == want.tolist(), f"{shape}: sequence {s}: {got.tolist()} vs {want.tolist()}"
    for m in (whole, a, mid, c):
        m.close()


_MAILBOX_PEER = r"""
import os, sys, json, numpy as np
root = sys.argv[1]; sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import llama_swift_amd as L
import synth
path, n_layer, n_steps = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
b = L.Model(path, n_ctx=64, layer_begin=1, layer_end=n_layer, n_seq=1)
b_hid, _, hh, _ = b.stage_mailbox(0)
print(json.dumps({"hidden_handle": hh.hex()}), flush=True)
msg = json.loads(sys.stdin.readline())                          # the first stage's token inbox + the prompt's hidden rows
b.stage_mailbox_connect(0, token_handle=bytes.fromhex(msg["token_handle"]))
hid = torch.tensor(msg["hidden"], dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
lg = b.eval_stage(0, n_tokens=msg["n_prompt"], hidden_in=hid.data_ptr(), want_logits=True)
first = int(np.argmax(lg))
print(json.dumps({"first": first}), flush=True)
sys.stdin.readline()                                            # "go": the first stage is bound
b.stage_bind(0, msg["n_prompt"])
print(json.dumps({"bound": True}), flush=True)                  # binding clears this stage's inbox: the first stage steps only after this
for _ in range(n_steps):
    b.stage_step(0, 8, 0)
try:
    n, pos, got = b.stage_trace(0, n_steps)
    print(json.dumps({"n": n, "pos": pos, "tokens": [int(t) for t in got]}), flush=True)
except L.LlamaHipError as e:
    print(json.dumps({"error": str(e), "code": e.code}), flush=True)
b.close()
"""


@pytest.mark.gpu
def test_device_side_mailboxes_two_processes_over_hip_ipc(L, tmp_path):
    """The same hand-off between two PROCESSES on one GPU: each stage's inbox is exported as a hipIpcMemHandle_t and opened by its
    neighbour -- what the stages of the 8-GPU pipeline do at bootstrap (the handles travel over the process group once; no collective
    per token after that).  The second stage runs in a child process; tokens must be those of the whole model."""
    import subprocess
    import sys
    import json
    hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=4)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=9))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n_steps = 6
    child = subprocess.Popen([sys.executable, "-c", _MAILBOX_PEER, root, path, "4", str(n_steps)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    try:
        a = L.Model(path, n_ctx=64, layer_begin=0, layer_end=1, n_seq=1)
        whole = L.Model(path, n_ctx=64)
        prompt = synth.synth_prompt(9, hp.n_vocab, seed=1)
        import torch
        h = torch.empty(len(prompt) * hp.n_embd, dtype=torch.float32, device="cuda")
        a.eval_stage(0, tokens=prompt, hidden_out=h.data_ptr())          # the prompt's residual rows after layer 0
        torch.cuda.synchronize()
        hid = h.cpu().numpy()
        _, a_tok, _, th = a.stage_mailbox(0)
        peer = json.loads(child.stdout.readline())
        a.stage_mailbox_connect(0, next_hidden_handle=bytes.fromhex(peer["hidden_handle"]))
        child.stdin.write(json.dumps({"token_handle": th.hex(), "hidden": [float(x) for x in hid], "n_prompt": len(prompt)}) + "\n"); child.stdin.flush()
        first = json.loads(child.stdout.readline())["first"]
        assert first == int(np.argmax(whole.eval(prompt, 0, 8)))
        tokbuf = torch.full((1,), first, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        a.stage_bind(0, len(prompt), token_in=tokbuf.data_ptr())
        child.stdin.write("go\n"); child.stdin.flush()
        assert json.loads(child.stdout.readline()).get("bound") is True
        for _ in range(n_steps):
            a.stage_step(0, 8, 0)
        res = json.loads(child.stdout.readline())
        assert "error" not in res, res
        assert a.stage_trace(0, 0)[:2] == (n_steps, len(prompt) + n_steps)
        whole.eval(prompt, 0, 8)
        want = whole.decode_greedy(first, len(prompt), n_steps, 8)
        assert res["tokens"] == want.tolist(), (res, want.tolist())
        a.close(); whole.close()
    finally:
        try:
            child.stdin.close()
        except Exception:
            pass
        child.wait(timeout=120)
        err = child.stderr.read()
        assert child.returncode == 0, err[-2000:]


@pytest.mark.gpu
def test_device_side_mailbox_lost_row_is_an_error_not_a_hang(tmp_path):
    """A row that never arrives (LLAMAHIP_HANDOFF_FAULT_TEST=3: the producing stage publishes a tag nobody waits for, polls shortened)
    raises the consumer's sticky fault word; llamahip_stage_trace returns PredictionFailed."""
    import subprocess
    import sys
    hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=4)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=9))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, time, numpy as np, torch\n"
        "import llama_swift_amd as L\n"
        "p = sys.argv[1]\n"
        "a = L.Model(p, n_ctx=64, layer_begin=0, layer_end=1); b = L.Model(p, n_ctx=64, layer_begin=1, layer_end=4)\n"
        "_, a_tok, _, _ = a.stage_mailbox(0); b_hid, _, _, _ = b.stage_mailbox(0)\n"
        "a.stage_mailbox_connect(0, next_hidden_ptr=b_hid); b.stage_mailbox_connect(0, token_ptr=a_tok)\n"
        "tok = torch.ones(1, dtype=torch.int32, device='cuda'); torch.cuda.synchronize()\n"
        "a.stage_bind(0, 0, token_in=tok.data_ptr()); b.stage_bind(0, 0)\n"
        "t0 = time.time()\n"
        "a.stage_step(0, 8, 0); b.stage_step(0, 8, 0)\n"
        "try:\n"
        "    b.stage_trace(0, 1); print('NO ERROR')\n"
        "except L.LlamaHipError as e:\n"
        "    print('ERR', e.code, str(e)); print('SECONDS', time.time() - t0)\n"
    )
    r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, LLAMAHIP_HANDOFF_FAULT_TEST="3", PYTHONPATH=root),
                       capture_output=True, text=True, cwd=root, timeout=300)
    assert "ERR -1001" in r.stdout and "hand-off" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert float(r.stdout.split("SECONDS")[1].split()[0]) < 30.0, r.stdout


@pytest.mark.gpu
def test_device_side_mailbox_silent_peer_costs_seconds_with_the_production_poll_bounds(tmp_path):
    """No fault injection, no shortened polls: a consumer stage whose producer never steps, and a first stage whose token never comes
    back, raise the fault word and return PredictionFailed after seconds (measured 5.1 / 4.7 s on one GPU) -- the price the N > 1
    bench's one-token handshake pays once before all ranks fall back to the RCCL hand-off (tools/mailbox_silence_probe.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mailbox_silence_probe.py"), str(tmp_path / "m.bin")],
                       env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, cwd=root, timeout=400)
    out = r.stdout
    assert out.count("ERR -1001") == 2 and "NO ERROR" not in out, out[-2000:] + r.stderr[-2000:]
    secs = [float(l.split()[-1]) for l in out.splitlines() if l.startswith("SECONDS")]
    assert len(secs) == 2 and max(secs) < 90.0, out


@pytest.mark.gpu
@pytest.mark.parametrize("S,shape,nth", [(2, "d512", 8), (4, "d256", 3), (8, "d512", 5), (16, "d256", 8), (5, "7b_width", 8), (9, "7b_width", 5), (3, "d512", 8),
                                          (4, "13b_width", 8), (8, "13b_width", 5), (4, "65b_width", 5), (8, "65b_width", 8)])
def test_batched_set_steps_equal_one_eval_per_sequence_on_the_oracle(L, oracle, tmp_path, S, shape, nth):
    """llamahip_stage_step_set: ONE decode step for S sequences at DIFFERENT positions (the weights are streamed once per step for
    all of them).  Every sequence's picked tokens, the logits of its last step and the KV rows of every layer must be bit for bit
    what one llama_eval per token and sequence gives on the oracle (.mm:510-735 row by row; the V*P key split of each row's own
    eval, ggml.c:5459-5480, n_threads 8 / 3 / 5).  Sets and single steps are interchangeable on the same slots (a few single
    steps in between, then a smaller set), captured graphs replay as the positions grow."""
    import torch
    # (the 13B / 65B widths: n_embd 5120 / 8192, 40 / 64 heads, n_ff 13824 / 22016, two- / eight-part files -- the shapes the 8-GPU
    #  pipeline's default schedule steps in sets of 4, .mm:33-38)
    kw = {"d512": dict(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=3), "d256": dict(n_vocab=96, n_embd=256, n_mult=64, n_head=2, n_layer=2),
          "7b_width": dict(n_vocab=512, n_embd=4096, n_mult=256, n_head=32, n_layer=2),
          "13b_width": dict(n_vocab=512, n_embd=5120, n_mult=256, n_head=40, n_layer=2, parts=2),
          "65b_width": dict(n_vocab=512, n_embd=8192, n_mult=256, n_head=64, n_layer=2, parts=8)}[shape]
    hp = synth.HParams(**{k: v for k, v in kw.items() if k != "parts"})
    path = str(tmp_path / "m.bin")
    paths_before = L.gemm_paths()
    if shape.endswith("_width"):
        from conftest import synth_tool
        path = synth_tool(tmp_path / "m.bin", seed=23, **kw)
    else:
        synth.write_model(path, hp, synth.random_tensors(hp, seed=23 + S))
    n_ctx = 96
    prompts = [synth.synth_prompt(3 + (5 * s) % 23, hp.n_vocab, seed=60 + s) for s in range(S)]      # positions 3 .. 25, all different mod 23
    K1, K2, K3 = 5, 2, 4                                       # set steps | single steps of every slot | steps of a smaller set
    with L.Model(path, n_ctx=n_ctx, n_seq=S) as gm:
        toks = []
        for s in range(S):
            gm.set_seq(s)
            toks.append(int(np.argmax(gm.eval(prompts[s], 0, nth))))
        gm.set_seq(0)
        bufs = [torch.tensor([toks[s]], dtype=torch.int32, device="cuda") for s in range(S)]
        for s in range(S):
            gm.stage_bind(s, len(prompts[s]), token_in=bufs[s].data_ptr(), token_out=bufs[s].data_ptr())
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(K1):
            gm.stage_step_set(list(range(S)), nth, st)
        last_full = [gm.stage_logits(s) for s in range(S)]
        for _ in range(K2):
            for s in range(S):
                gm.stage_step(s, nth, st)
        sub = list(range(S - 1, 0, -2))                        # a smaller set in another order (its own graph and descriptor)
        if len(sub) >= 2:
            for _ in range(K3):
                gm.stage_step_set(sub, nth, st)
        last_sub = [gm.stage_logits(i) for i in range(len(sub))] if len(sub) >= 2 else []
        for s in range(S):
            n_steps = K1 + K2 + (K3 if (s in sub and len(sub) >= 2) else 0)
            n, pos, got = gm.stage_trace(s, n_steps)
            assert n == n_steps and pos == len(prompts[s]) + n_steps
            om = oracle.load(path, n_ctx)
            lo = om.eval(prompts[s], 0, nth)["logits"]
            t = int(np.argmax(lo))
            assert t == toks[s]
            want = []
            for i in range(n_steps):
                lo = om.eval(np.array([t], np.int32), len(prompts[s]) + i, nth)["logits"]
                if i == K1 - 1:
                    assert same(last_full[s], lo), f"sequence {s}: logits of the last full-set step"
                t = int(np.argmax(lo)); want.append(t)
            assert got.tolist() == want, f"sequence {s}: {got.tolist()} vs {want}"
            if s in sub and len(sub) >= 2:
                assert same(last_sub[sub.index(s)], lo), f"sequence {s}: logits of the last sub-set step"
            gm.set_seq(s)
            for il in range(hp.n_layer):
                gk, gv = gm.kv(il, len(prompts[s]) + n_steps)
                ok, ov = om.kv(il, len(prompts[s]) + n_steps)
                assert same(gk, ok) and same(gv, ov), f"sequence {s}: KV cache layer {il}"
            om.close()
        with pytest.raises(L.LlamaHipError, match="twice"):
            gm.stage_step_set([0, 1, 0], nth, st)
    # the set steps (and the prompts' short evals) ran on the few-row kernel (k_gemv_set), none of their mat-muls on a generic fall-back
    after = L.gemm_paths()
    assert after["set"] > paths_before["set"] and after["lds"] == paths_before["lds"] and after["rows"] == paths_before["rows"], (paths_before, after)


@pytest.mark.gpu
def test_batched_set_steps_through_two_stage_handles(L, tmp_path):
    """Set steps on layer-range handles: the set's residual rows are gathered from / scattered to the slots' hidden buffers; two
    stages on one GPU chained on one stream equal the whole-model greedy loop for every sequence."""
    import torch
    hp = synth.HParams(n_vocab=160, n_embd=512, n_mult=256, n_head=4, n_layer=4)
    path = str(tmp_path / "m.bin")
    synth.write_model(path, hp, synth.random_tensors(hp, seed=31))
    S, K = 4, 6
    prompts = [synth.synth_prompt(4 + 3 * s, hp.n_vocab, seed=80 + s) for s in range(S)]
    with L.Model(path, n_ctx=64) as whole:
        firsts, wants = [], []
        for s in range(S):
            f = int(np.argmax(whole.eval(prompts[s], 0, 8)))
            firsts.append(f); wants.append(whole.decode_greedy(f, len(prompts[s]), K, 8).tolist())
    a = L.Model(path, n_ctx=64, layer_begin=0, layer_end=2, n_seq=S)
    b = L.Model(path, n_ctx=64, layer_begin=2, layer_end=4, n_seq=S)
    hid = [torch.zeros(hp.n_embd, dtype=torch.float32, device="cuda") for _ in range(S)]
    tok = [torch.tensor([firsts[s]], dtype=torch.int32, device="cuda") for s in range(S)]
    for s in range(S):
        a.set_seq(s); b.set_seq(s)
        h = torch.zeros(len(prompts[s]) * hp.n_embd, dtype=torch.float32, device="cuda")
        a.eval_stage(0, tokens=prompts[s], hidden_out=h.data_ptr())
        b.eval_stage(0, n_tokens=len(prompts[s]), hidden_in=h.data_ptr())
        a.stage_bind(s, len(prompts[s]), token_in=tok[s].data_ptr(), hidden_out=hid[s].data_ptr())
        b.stage_bind(s, len(prompts[s]), hidden_in=hid[s].data_ptr(), token_out=tok[s].data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(K):
        a.stage_step_set(list(range(S)), 8, st)
        b.stage_step_set(list(range(S)), 8, st)
    for s in range(S):
        n, pos, got = b.stage_trace(s, K)
        assert n == K and got.tolist() == wants[s], f"sequence {s}"
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,S,nth", [("65b_width", 4, 8), ("13b_width", 8, 5)])
def test_batched_set_steps_through_two_stage_handles_vs_the_oracle(L, oracle, tmp_path, shape, S, nth):
    """The 8-GPU pipeline's default schedule in miniature, against the CPU path and not against the whole-model HIP path: a 2-layer model
    of the 65B / 13B width (eight- / two-part file) split into two one-layer stage handles, S sequences at different positions stepped
    as ONE set per stage and step (llamahip_stage_step_set; rows gathered from / scattered to the slots' hidden buffers).  Every
    sequence's tokens, the logits of its last step and the KV rows of both layers equal one llama_eval per token and sequence on the
    oracle (.mm:563-705 row by row, the V*P key split of each row's own eval, ggml.c:5459-5480)."""
    import torch
    from conftest import synth_tool
    kw = {"13b_width": dict(n_vocab=512, n_embd=5120, n_mult=256, n_head=40, n_layer=2, parts=2),
          "65b_width": dict(n_vocab=512, n_embd=8192, n_mult=256, n_head=64, n_layer=2, parts=8)}[shape]
    hp = synth.HParams(**{k: v for k, v in kw.items() if k != "parts"})
    path = synth_tool(tmp_path / "m.bin", seed=37, **kw)
    n_ctx, K = 64, 6
    prompts = [synth.synth_prompt(3 + (7 * s) % 19, hp.n_vocab, seed=90 + s) for s in range(S)]
    a = L.Model(path, n_ctx=n_ctx, layer_begin=0, layer_end=1, n_seq=S)
    b = L.Model(path, n_ctx=n_ctx, layer_begin=1, layer_end=2, n_seq=S)
    try:
        firsts = []
        hid = [torch.zeros(hp.n_embd, dtype=torch.float32, device="cuda") for _ in range(S)]
        for s in range(S):
            a.set_seq(s); b.set_seq(s)
            h = torch.zeros(len(prompts[s]) * hp.n_embd, dtype=torch.float32, device="cuda")
            a.eval_stage(0, tokens=prompts[s], hidden_out=h.data_ptr(), n_threads=nth)
            lg = b.eval_stage(0, n_tokens=len(prompts[s]), hidden_in=h.data_ptr(), want_logits=True, n_threads=nth)
            firsts.append(int(np.argmax(lg)))
        tok = [torch.tensor([firsts[s]], dtype=torch.int32, device="cuda") for s in range(S)]
        for s in range(S):
            a.stage_bind(s, len(prompts[s]), token_in=tok[s].data_ptr(), hidden_out=hid[s].data_ptr())
            b.stage_bind(s, len(prompts[s]), hidden_in=hid[s].data_ptr(), token_out=tok[s].data_ptr())
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(K):
            a.stage_step_set(list(range(S)), nth, st)
            b.stage_step_set(list(range(S)), nth, st)
        last = [b.stage_logits(s) for s in range(S)]
        for s in range(S):
            n, pos, got = b.stage_trace(s, K)
            assert n == K and pos == len(prompts[s]) + K
            om = oracle.load(path, n_ctx)
            lo = om.eval(prompts[s], 0, nth)["logits"]
            t = int(np.argmax(lo))
            assert t == firsts[s], f"sequence {s}: prompt pick"
            want = []
            for i in range(K):
                lo = om.eval(np.array([t], np.int32), len(prompts[s]) + i, nth)["logits"]
                t = int(np.argmax(lo)); want.append(t)
            assert got.tolist() == want, f"sequence {s}: {got.tolist()} vs {want}"
            assert same(last[s], lo), f"sequence {s}: logits of the last step"
            for il, stage in ((0, a), (1, b)):
                stage.set_seq(s)
                gk, gv = stage.kv(il, len(prompts[s]) + K)
                ok, ov = om.kv(il, len(prompts[s]) + K)
                assert same(gk, ok) and same(gv, ov), f"sequence {s}: KV cache layer {il}"
            om.close()
    finally:
        a.close(); b.close()

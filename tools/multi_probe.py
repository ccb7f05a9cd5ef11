#!/usr/bin/env python3
"""llamahip_decode_greedy_multi on ONE GPU (measurement tooling): aggregate tokens/s of n sequences decoded together on a plain handle (stages = 1) and
on pipeline handles with every stage on device 0 (the groups of sequences pipelined over the stages: their sets overlap on the shared GPU).  Sequence 0
decodes the bench prompt and must reproduce the single-stream tokens.
usage: multi_probe.py [model = 7B] [stages:sequences,... = 1:8,1:16,2:8,2:16,2:32,4:32,8:64]"""
import os, sys, json, argparse
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench, llama_swift_amd as L
model = sys.argv[1] if len(sys.argv) > 1 else "7B"
cases = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[2] if len(sys.argv) > 2 else "1:8,1:16,2:8,2:16,2:32,4:32,8:64").split(",")]
cfg = bench.MODELS[model]; path = bench.model_path(model, cfg, 20230312)
args = argparse.Namespace(n_ctx=512, threads=8)
with L.Model(path, n_ctx=512) as m:
    p = bench.PROMPT % cfg["n_vocab"]; p[0] = 1
    f = int(np.argmax(m.eval(p, 0, 8))); tr = m.decode_greedy(f, len(p), 200, 8).tolist()
for st, sq in cases:
    print(json.dumps(bench.inprocess_pipeline_multi(args, cfg, path, st, sq, tr)), flush=True)

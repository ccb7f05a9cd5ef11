#!/bin/bash
# Round-4 pass C: pipeline GPU tests (set steps, mailboxes), the N > 1 bench path forced onto one GPU (one rank, set mode), and two
# ranks on one GPU over gloo (set schedule across two processes).
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_pipeline.py -x -q -m gpu --durations=5 > $O/r04c_pipeline_tests.txt 2>&1; tail -12 $O/r04c_pipeline_tests.txt
LLAMAHIP_FORCE_PIPELINE=1 LLAMAHIP_BENCH_65B=0 timeout 600 python bench.py --gpus 1 --steps 64 > $O/r04c_forced_pipeline_1rank.json 2> $O/r04c_forced_pipeline_1rank.log; tail -3 $O/r04c_forced_pipeline_1rank.log; python - <<PY
import json
d = json.load(open("$O/r04c_forced_pipeline_1rank.json"))
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["config"]["hand_off"], d["parity"], d["roofline"].get("kernel"), d["roofline"].get("frac"))
PY
LLAMAHIP_PIPE_ONE_GPU=1 LLAMAHIP_PIPE_BACKEND=gloo LLAMAHIP_BENCH_65B=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 64 > $O/r04c_2ranks_one_gpu_sets.json 2> $O/r04c_2ranks_one_gpu_sets.log; tail -5 $O/r04c_2ranks_one_gpu_sets.log; python - <<PY
import json
d = json.load(open("$O/r04c_2ranks_one_gpu_sets.json"))
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["config"]["hand_off"], d["parity"], d["single_stream"], d["roofline"].get("kernel"), d["roofline"].get("frac"))
PY

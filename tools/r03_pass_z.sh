#!/bin/bash
# Round-3 GPU pass z: llamahip_eval_chunks (the reference's 9-token prompt loop in one chunk-exact pass): parity, runner events, speed
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_pass or runner or prompt_continuation or short_chunks or long_prompt" > $O/r03z_quick.txt 2>&1; tail -5 $O/r03z_quick.txt
timeout 600 python tools/prefill_probe.py > $O/r03z_prefill_probe.txt 2>&1; cat $O/r03z_prefill_probe.txt | tail -6
timeout 300 python tools/runner_probe.py > $O/r03z_runner_probe.txt 2>&1; tail -5 $O/r03z_runner_probe.txt

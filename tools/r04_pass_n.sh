#!/bin/bash
# Round-4 pass N: w1|w3 in half-block workgroups (EPI_SILU_QAH): parity + A/B against the 8-wave block workgroups
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wider_models or 7b_logits or greedy_trace_128 or ragged_contexts or dc_offset or handoff_timeout" > $O/r04n_pytest.txt 2>&1; tail -4 $O/r04n_pytest.txt
timeout 600 python -m pytest tests/test_pipeline.py -x -q -m gpu -k "batched_set or stream_ordered" >> $O/r04n_pytest.txt 2>&1; tail -3 $O/r04n_pytest.txt
cat > /tmp/variants.txt <<V
w13_blocks|LLAMAHIP_NO_W13_HALF=1
w13_halves|LLAMAHIP_X=1
V
PROF=1 STEPS=64 AT=8,256 FILTER='k_gemv<4, \|k_qkv' tools/decode_ab.sh /tmp/variants.txt > $O/r04n_w13_half_ab.txt 2>&1; cat $O/r04n_w13_half_ab.txt

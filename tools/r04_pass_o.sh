#!/bin/bash
# Round-4 pass O: k_qkv_attn's mat-vec role in workgroups of 3 row-groups (512 = two per CU at 7B): parity + A/B against 384 x 4
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wider_models or 7b_logits or greedy_trace_128 or ragged_contexts or dc_offset or handoff_timeout or thread_splits" > $O/r04o_pytest.txt 2>&1; tail -4 $O/r04o_pytest.txt
cat > /tmp/variants.txt <<V
qkv_4rg|LLAMAHIP_QKV_RG3=0
qkv_3rg|LLAMAHIP_X=1
V
PROF=1 STEPS=64 AT=8,256,440 FILTER='k_qkv' tools/decode_ab.sh /tmp/variants.txt > $O/r04o_qkv_rg3_ab.txt 2>&1; cat $O/r04o_qkv_rg3_ab.txt

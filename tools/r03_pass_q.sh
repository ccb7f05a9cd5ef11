#!/bin/bash
# (as run at the time: LLAMAHIP_PV_STAGE / LLAMAHIP_ATTNQ_PF were tuning switches of that build; removed with the variants that lost)
# Round-3 GPU pass q: long-context decode attention (k_dec_pv_stream) -- parity subset, tokens/s by context for the schedules, per-kernel times
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(fallback_paths and (switch2 or switch3 or switch4)) or width_2048" > $O/r03q_quick.txt 2>&1; tail -3 $O/r03q_quick.txt
cat > /tmp/v7.txt <<EOV
fused_or_two|LLAMAHIP_ATTN_LONG_FROM=-1
stream4|LLAMAHIP_ATTN_LONG_FROM=0
stream2|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_STAGE=2
two_launch|LLAMAHIP_ATTN_LONG_FROM=-1 LLAMAHIP_NO_ATTN_X=1
EOV
N_CTX=2560 STEPS=64 AT=128,520,800,1024,1536,2048 timeout 1200 bash tools/decode_ab.sh /tmp/v7.txt > $O/r03q_7b.txt 2>&1
cat $O/r03q_7b.txt
cat > /tmp/v13.txt <<EOV
fused|LLAMAHIP_ATTN_LONG_FROM=-1
stream4|LLAMAHIP_ATTN_LONG_FROM=0
EOV
MODEL=13B N_CTX=2560 STEPS=64 AT=128,400,800,2048 timeout 1200 bash tools/decode_ab.sh /tmp/v13.txt > $O/r03q_13b.txt 2>&1
cat $O/r03q_13b.txt
cat > /tmp/v1.txt <<EOV
stream4_at_2048|LLAMAHIP_ATTN_LONG_FROM=0
fused_at_2048|LLAMAHIP_ATTN_LONG_FROM=-1
EOV
PROF=1 KEEP=1 N_CTX=2560 STEPS=64 AT=8 PROF_AT=2048 FILTER='k_gemv\|k_qkv\|k_dec\|k_embed\|k_argmax' timeout 900 bash tools/decode_ab.sh /tmp/v1.txt > $O/r03q_prof.txt 2>&1
cat $O/r03q_prof.txt

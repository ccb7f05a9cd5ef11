#!/bin/bash
# Round-3 GPU pass w: tokens/s by context for the three decode attention schedules, 7B / 13B / 65B (-> profiles/r03_attn_by_context.txt)
O=gpurun_out; mkdir -p $O
cat > /tmp/vs.txt <<EOV
fused|LLAMAHIP_ATTN_TWO_FROM=-1 LLAMAHIP_ATTN_LONG_FROM=-1
two|LLAMAHIP_ATTN_TWO_FROM=0 LLAMAHIP_ATTN_LONG_FROM=-1
stream|LLAMAHIP_ATTN_LONG_FROM=0
default|LLAMAHIP_X=1
EOV
N_CTX=2560 STEPS=48 AT=64,256,448,640,896,1152,1408,1792,2304 timeout 1200 bash tools/decode_ab.sh /tmp/vs.txt > $O/r03w_7b.txt 2>&1
cat $O/r03w_7b.txt
MODEL=13B N_CTX=2560 STEPS=48 AT=64,256,448,640,896,1152,1408,1792,2304 timeout 1200 bash tools/decode_ab.sh /tmp/vs.txt > $O/r03w_13b.txt 2>&1
cat $O/r03w_13b.txt
MODEL=65B N_CTX=2560 STEPS=32 AT=64,448,896,1408,2304 timeout 1500 bash tools/decode_ab.sh /tmp/vs.txt > $O/r03w_65b.txt 2>&1
cat $O/r03w_65b.txt

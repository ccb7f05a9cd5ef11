#!/bin/bash
# Round-4: tokens/s by context for the three decode attention schedules, 7B (-> profiles/r04_attn_by_context.txt)
O=gpurun_out; mkdir -p $O
bash tools/ensure_7b.sh
cat > /tmp/vs.txt <<EOV
fused|LLAMAHIP_ATTN_TWO_FROM=-1 LLAMAHIP_ATTN_LONG_FROM=-1
two|LLAMAHIP_ATTN_TWO_FROM=0 LLAMAHIP_ATTN_LONG_FROM=-1
stream|LLAMAHIP_ATTN_LONG_FROM=0
default|LLAMAHIP_X=1
EOV
N_CTX=2560 STEPS=48 AT=64,256,448,640,896,1152,1408,1792,2304 timeout 270 bash tools/decode_ab.sh /tmp/vs.txt > $O/r04ctx_7b.txt 2>&1
cat $O/r04ctx_7b.txt

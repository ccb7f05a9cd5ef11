#!/bin/bash
# Round-3 GPU pass u: k_dec_pv_stream asking for V only after the score row arrived; issue barrier in the decode mat-vecs (variant b)
O=gpurun_out; mkdir -p $O
echo "== LLAMAHIP_PV_SPLIT=2 at 2048" > $O/r03u_timeline.txt
LLAMAHIP_PV_SPLIT=2 timeout 600 python tools/pv_stream_timeline.py 2048 2 2>&1 | tail -28 >> $O/r03u_timeline.txt
cat $O/r03u_timeline.txt
cat > /tmp/v7.txt <<EOV
fused|LLAMAHIP_ATTN_LONG_FROM=-1
fused_issue_barrier|LLAMAHIP_ATTN_LONG_FROM=-1 LLAMAHIP_LIB=libllamahip_b.so
split2|LLAMAHIP_ATTN_LONG_FROM=0
split2_sr64|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_STAGE_ROWS=64
split1|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_SPLIT=1
EOV
N_CTX=2560 STEPS=64 AT=128,520,800,1024,1536,2048 timeout 1500 bash tools/decode_ab.sh /tmp/v7.txt > $O/r03u_7b.txt 2>&1
cat $O/r03u_7b.txt
cat > /tmp/v13.txt <<EOV
fused|LLAMAHIP_ATTN_LONG_FROM=-1
fused_issue_barrier|LLAMAHIP_ATTN_LONG_FROM=-1 LLAMAHIP_LIB=libllamahip_b.so
split1|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_SPLIT=1
split2|LLAMAHIP_ATTN_LONG_FROM=0
EOV
MODEL=13B N_CTX=2560 STEPS=64 AT=128,400,800,2048 timeout 1200 bash tools/decode_ab.sh /tmp/v13.txt > $O/r03u_13b.txt 2>&1
cat $O/r03u_13b.txt

#!/bin/bash
# Round-3 GPU pass t: k_dec_pv_stream / k_dec_scores with the small loads ordered before the bulk loads; stage sizes
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fallback_paths and (switch1 or switch3)" > $O/r03t_quick.txt 2>&1; tail -3 $O/r03t_quick.txt
echo "== LLAMAHIP_PV_SPLIT=2 at 2048" > $O/r03t_timeline.txt
LLAMAHIP_PV_SPLIT=2 timeout 600 python tools/pv_stream_timeline.py 2048 2 2>&1 | tail -28 >> $O/r03t_timeline.txt
cat $O/r03t_timeline.txt
cat > /tmp/v7.txt <<EOV
fused|LLAMAHIP_ATTN_LONG_FROM=-1
split2|LLAMAHIP_ATTN_LONG_FROM=0
split2_sr64|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_STAGE_ROWS=64
split2_sr32|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_STAGE_ROWS=32
split1|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_SPLIT=1
split1_sr32|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_SPLIT=1 LLAMAHIP_PV_STAGE_ROWS=32
two_launch|LLAMAHIP_ATTN_LONG_FROM=-1 LLAMAHIP_NO_ATTN_X=1
EOV
N_CTX=2560 STEPS=64 AT=128,520,800,1024,1536,2048 timeout 1500 bash tools/decode_ab.sh /tmp/v7.txt > $O/r03t_7b.txt 2>&1
cat $O/r03t_7b.txt
cat > /tmp/v13.txt <<EOV
fused|LLAMAHIP_ATTN_LONG_FROM=-1
split1|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_SPLIT=1
split1_sr32|LLAMAHIP_ATTN_LONG_FROM=0 LLAMAHIP_PV_SPLIT=1 LLAMAHIP_PV_STAGE_ROWS=32
EOV
MODEL=13B N_CTX=2560 STEPS=64 AT=128,400,800,2048 timeout 1200 bash tools/decode_ab.sh /tmp/v13.txt > $O/r03t_13b.txt 2>&1
cat $O/r03t_13b.txt
cat > /tmp/v1.txt <<EOV
split2_at_2048|LLAMAHIP_ATTN_LONG_FROM=0
EOV
PROF=1 KEEP=1 N_CTX=2560 STEPS=64 AT=8 PROF_AT=2048 FILTER='k_gemv\|k_qkv\|k_dec\|k_embed\|k_argmax' timeout 900 bash tools/decode_ab.sh /tmp/v1.txt > $O/r03t_prof.txt 2>&1
cat $O/r03t_prof.txt

#!/bin/bash
# Round-3 GPU pass i: L2 warm-up of wo / w1|w3 by the exiting mat-vec workgroups of k_qkv_attn (prefetch_tail)
O=gpurun_out; mkdir -p $O; R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "thread_splits" > $O/r03i_quick.txt 2>&1; tail -2 $O/r03i_quick.txt
cat > /tmp/v2.txt <<EOV
base|LLAMAHIP_NO_TAIL_PREFETCH=1
tail_default|LLAMAHIP_X=1
wo_only|LLAMAHIP_PF_W13_TILES=0
w13_2|LLAMAHIP_PF_W13_TILES=2
w13_4|LLAMAHIP_PF_W13_TILES=4
w13_6|LLAMAHIP_PF_W13_TILES=6
w13_8|LLAMAHIP_PF_W13_TILES=8
w13_only3|LLAMAHIP_PF_WO_TILES=0
EOV
STEPS=64 AT=8,256,440 timeout 1500 bash tools/decode_ab.sh /tmp/v2.txt > $O/r03i_ab.txt 2>&1
cat $O/r03i_ab.txt
cat > /tmp/v1.txt <<EOV
tail_default|LLAMAHIP_X=1
EOV
PROF=1 KEEP=1 STEPS=64 AT=8 PROF_AT=128 FILTER='k_gemv\|k_qkv\|k_embed\|k_argmax' timeout 900 bash tools/decode_ab.sh /tmp/v1.txt > $O/r03i_ab_prof.txt 2>&1
cat $O/r03i_ab_prof.txt

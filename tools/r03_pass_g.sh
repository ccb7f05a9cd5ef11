#!/bin/bash
# Round-3 GPU pass g: the L2 prefetcher with its loads actually in flight (16 per wave, branch-free)
O=gpurun_out; mkdir -p $O; R=$PWD
cat > /tmp/v2.txt <<EOV
base|LLAMAHIP_NO_PREFETCH=1
pf16|LLAMAHIP_PF_BUDGET_MB=16
pf8|LLAMAHIP_PF_BUDGET_MB=8
pf24|LLAMAHIP_PF_BUDGET_MB=24
pf32|LLAMAHIP_PF_BUDGET_MB=32
pf48|LLAMAHIP_PF_BUDGET_MB=48
pf16_w64|LLAMAHIP_PF_WGS=64
pf16_w32|LLAMAHIP_PF_WGS=32
pf16_w256|LLAMAHIP_PF_WGS=256
pf16_xcc_wrong|LLAMAHIP_PF_XCC0=3
pf_nothrottle|LLAMAHIP_PF_MODE=2
EOV
STEPS=64 AT=8,256 timeout 1500 bash tools/decode_ab.sh /tmp/v2.txt > $O/r03g_ab.txt 2>&1
cat $O/r03g_ab.txt
cat > /tmp/v1.txt <<EOV
pf16|LLAMAHIP_PF_BUDGET_MB=16
EOV
PROF=1 KEEP=1 STEPS=64 AT=8,256,440 PROF_AT=128 FILTER='k_gemv\|k_qkv\|k_embed\|k_argmax\|k_prefetch' timeout 900 bash tools/decode_ab.sh /tmp/v1.txt > $O/r03g_ab_prof.txt 2>&1
cat $O/r03g_ab_prof.txt

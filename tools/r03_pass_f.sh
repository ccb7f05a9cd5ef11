#!/bin/bash
# Round-3 GPU pass f: what does the L2 prefetcher cost -- its polling, its traffic, or its presence?
O=gpurun_out; mkdir -p $O; R=$PWD
cat > /tmp/v2.txt <<EOV
base|LLAMAHIP_NO_PREFETCH=1
pf_pollonly|LLAMAHIP_PF_MODE=1
pf_pollonly_nap8|LLAMAHIP_PF_MODE=1 LLAMAHIP_PF_NAP=8
pf_pollonly_w8|LLAMAHIP_PF_MODE=1 LLAMAHIP_PF_WGS=8
pf_nothrottle|LLAMAHIP_PF_MODE=2
pf_nothrottle_w8|LLAMAHIP_PF_MODE=2 LLAMAHIP_PF_WGS=8
pf_nap8|LLAMAHIP_PF_NAP=8
pf_nap8_b32|LLAMAHIP_PF_NAP=8 LLAMAHIP_PF_BUDGET_MB=32
pf_nap4_w256|LLAMAHIP_PF_NAP=4 LLAMAHIP_PF_WGS=256
pf_w8|LLAMAHIP_PF_WGS=8
EOV
STEPS=64 AT=8,256 timeout 1500 bash tools/decode_ab.sh /tmp/v2.txt > $O/r03f_ab.txt 2>&1
cat $O/r03f_ab.txt

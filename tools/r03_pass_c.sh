#!/bin/bash
# Round-3 third GPU pass: the L2 run-ahead prefetcher beside the decode loop -- A/B over budget / workgroups / line size.
O=gpurun_out; mkdir -p $O; R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "thread_splits or greedy_trace_128" > $O/r03e_quick.txt 2>&1
tail -3 $O/r03e_quick.txt
cat > /tmp/v1.txt <<EOV
base|LLAMAHIP_NO_PREFETCH=1
pf16|LLAMAHIP_PF_BUDGET_MB=16
EOV
PROF=1 KEEP=1 STEPS=64 AT=8,256,440 PROF_AT=128 FILTER='k_gemv\|k_qkv\|k_embed\|k_argmax\|k_prefetch' timeout 900 bash tools/decode_ab.sh /tmp/v1.txt > $O/r03e_ab_prof.txt 2>&1
cat $O/r03e_ab_prof.txt
cat > /tmp/v2.txt <<EOV
pf4|LLAMAHIP_PF_BUDGET_MB=4
pf8|LLAMAHIP_PF_BUDGET_MB=8
pf12|LLAMAHIP_PF_BUDGET_MB=12
pf24|LLAMAHIP_PF_BUDGET_MB=24
pf32|LLAMAHIP_PF_BUDGET_MB=32
pf64|LLAMAHIP_PF_BUDGET_MB=64
pf16_w32|LLAMAHIP_PF_WGS=32
pf16_w64|LLAMAHIP_PF_WGS=64
pf16_w256|LLAMAHIP_PF_WGS=256
pf16_l64|LLAMAHIP_PF_LINE=64
pf16_l256|LLAMAHIP_PF_LINE=256
pf16_eager|PROBE_FLAGS=1
pf16_xcc_wrong|LLAMAHIP_PF_XCC0=3
EOV
STEPS=64 AT=8,256,440 timeout 1500 bash tools/decode_ab.sh /tmp/v2.txt > $O/r03e_ab.txt 2>&1
cat $O/r03e_ab.txt
rm -rf /tmp/tl_pf2
(cd /tmp && export TMPDIR=/tmp && env LLAMAHIP_WITH_TORCH=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_pf2 -o tl -- python $R/tools/decode_probe.py --steps 24 --at 128 --reps 1 > /tmp/tl_pf2.log 2>&1)
python tools/overlap_timeline.py /tmp/tl_pf2 --layers 3 > $O/r03e_timeline_pf.txt 2>&1
cat $O/r03e_timeline_pf.txt

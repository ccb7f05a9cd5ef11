#!/bin/bash
# Round-4 pass V: prompt attention: V*P column quarters, queries per launch
O=gpurun_out; mkdir -p $O
bash tools/ensure_7b.sh
{
echo "== default (V*P column halves, 512 queries per launch)"; timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== V*P column quarters   [LLAMAHIP_PV_NCB=1]"; LLAMAHIP_PV_NCB=1 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== 1024 queries per launch   [LLAMAHIP_ATTN_NB=1024]"; LLAMAHIP_ATTN_NB=1024 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== 2048 queries per launch   [LLAMAHIP_ATTN_NB=2048]"; LLAMAHIP_ATTN_NB=2048 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== 2048 queries per launch, quarters   [LLAMAHIP_ATTN_NB=2048 LLAMAHIP_PV_NCB=1]"; LLAMAHIP_ATTN_NB=2048 LLAMAHIP_PV_NCB=1 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== 256 queries per launch   [LLAMAHIP_ATTN_NB=256]"; LLAMAHIP_ATTN_NB=256 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
} > $O/r04v_attn_ab.txt 2>&1; cat $O/r04v_attn_ab.txt

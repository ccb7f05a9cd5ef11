#!/bin/bash
# Round-4 pass W: prompt scores at three waves per SIMD (groups of eight MFMAs): parity + A/B
O=gpurun_out; mkdir -p $O
bash tools/ensure_7b.sh
LLAMAHIP_SCORES3=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "2048_token_prefill or long_prompt or multipart or prompt_continuation" > $O/r04w_pytest.txt 2>&1; tail -3 $O/r04w_pytest.txt
{
echo "== scores: two waves per SIMD, 32 MFMAs then the tree"; timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== scores: three waves per SIMD, groups of eight   [LLAMAHIP_SCORES3=1]"; LLAMAHIP_SCORES3=1 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== again: two waves"; timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "== again: three waves"; LLAMAHIP_SCORES3=1 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
} > $O/r04w_scores3_ab.txt 2>&1; cat $O/r04w_scores3_ab.txt

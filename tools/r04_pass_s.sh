#!/bin/bash
# Round-4 pass S: k_gemm_mfma4 with one-byte weights (fp16 high bytes, v_perm_b32 operands): parity + A/B against k_gemm_mfma16
O=gpurun_out; mkdir -p $O
bash tools/ensure_7b.sh
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_core_prompt_gemm or 2048_token_prefill or long_prompt or multipart" > $O/r04s_pytest.txt 2>&1; tail -4 $O/r04s_pytest.txt
{
echo "== k_gemm_mfma16 (two waves per SIMD)   [LLAMAHIP_GEMM4=0]"; LLAMAHIP_GEMM4=0 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids
echo "== k_gemm_mfma4 (four waves per SIMD, wave = 16 x 32 outputs x 8 chains)"; timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids
} > $O/r04s_gemm4_ab.txt 2>&1; cat $O/r04s_gemm4_ab.txt

#!/bin/bash
# Round-4 pass Z: k_gemm_mfma2b (32x32x4_2b, three waves per SIMD, 32 x 96 workgroup tiles) against k_gemm_mfma4: parity + A/B
O=gpurun_out; mkdir -p $O
bash tools/ensure_7b.sh
LLAMAHIP_GEMM2B=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_core_prompt_gemm or 2048_token_prefill or long_prompt" > $O/r04z_pytest.txt 2>&1; tail -3 $O/r04z_pytest.txt
{
echo "== k_gemm_mfma4"; timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== k_gemm_mfma2b   [LLAMAHIP_GEMM2B=1]"; LLAMAHIP_GEMM2B=1 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
} > $O/r04z_gemm2b_ab.txt 2>&1; cat $O/r04z_gemm2b_ab.txt

#!/bin/bash
# Round-4 pass Q: k_gemm_mfma4 with packed FMAs: A/B (parity of the packed variant: gpurun_out/r04q_pytest.txt)
O=gpurun_out; mkdir -p $O
bash tools/ensure_7b.sh
{
echo "== k_gemm_mfma4, plain v_fma_f32   [LLAMAHIP_GEMM4_PK=0]"; LLAMAHIP_GEMM4_PK=0 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids
echo "== k_gemm_mfma4, v_pk_fma_f32 / v_pk_mul_f32"; timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids
} > $O/r04q_gemm4_pk_ab.txt 2>&1; cat $O/r04q_gemm4_pk_ab.txt

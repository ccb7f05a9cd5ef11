#!/bin/bash
# Round-4 pass P: k_gemm_mfma4 (four waves per SIMD, chain halves, DMA operands): layout probe, parity, A/B against k_gemm_mfma16
O=gpurun_out; mkdir -p $O
tools/mfma_layout_probe4 > $O/r04p_layout.txt 2>&1; cat $O/r04p_layout.txt
# pending from the previous change: w1|w3 workgroup shapes on the wider models
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fallback_paths and w13" > $O/r04p_pytest.txt 2>&1; tail -3 $O/r04p_pytest.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_core_prompt_gemm or 2048_token_prefill or long_prompt or multipart" >> $O/r04p_pytest.txt 2>&1; tail -4 $O/r04p_pytest.txt
{
echo "== k_gemm_mfma16 (two waves per SIMD)   [LLAMAHIP_GEMM4=0]"; LLAMAHIP_GEMM4=0 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids
echo "== k_gemm_mfma4 (four waves per SIMD)"; timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids
} > $O/r04p_gemm4_ab.txt 2>&1; cat $O/r04p_gemm4_ab.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pf1
LLAMAHIP_WITH_TORCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf1 -o p -- python $OLDPWD/tools/prefill_one.py 2048 2 > /tmp/pf1.log 2>&1
cd $OLDPWD
python tools/prof_summary.py $(find /tmp/pf1 -name "*kernel_stats.csv") "rocprofv3 --kernel-trace --stats -- python tools/prefill_one.py 2048 2   (k_gemm_mfma4)" > $O/r04p_prefill_2048_kernel_stats.txt; head -12 $O/r04p_prefill_2048_kernel_stats.txt

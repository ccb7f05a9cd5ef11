#!/bin/bash
# Round-4 pass U: prompt attention with more waves per SIMD (scores: 4 MFMAs at a time, <= 128 registers; V*P: column halves): parity + A/B
O=gpurun_out; mkdir -p $O
bash tools/ensure_7b.sh
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_core_prompt_gemm or 2048_token_prefill or long_prompt or multipart or chunk" > $O/r04u_pytest.txt 2>&1; tail -4 $O/r04u_pytest.txt
{
echo "== V*P: one wave per 128 columns   [LLAMAHIP_PV_NCB=4]"; LLAMAHIP_PV_NCB=4 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
echo "== V*P: column halves (default)"; timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
echo "== scores: 16384 waves per launch   [LLAMAHIP_SCORES_WAVES=16384]"; LLAMAHIP_SCORES_WAVES=16384 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
echo "== scores: 8192 waves per launch   [LLAMAHIP_SCORES_WAVES=8192]"; LLAMAHIP_SCORES_WAVES=8192 timeout 300 python tools/prefill_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
} > $O/r04u_attn_ab.txt 2>&1; cat $O/r04u_attn_ab.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pf1
LLAMAHIP_WITH_TORCH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf1 -o p -- python $R/tools/prefill_one.py 2048 2 > /tmp/pf1.log 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/pf1 -name "*kernel_stats.csv") "rocprofv3 --kernel-trace --stats -- python tools/prefill_one.py 2048 2" > $O/r04u_prefill_2048_kernel_stats.txt; head -10 $O/r04u_prefill_2048_kernel_stats.txt

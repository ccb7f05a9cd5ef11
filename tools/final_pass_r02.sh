#!/bin/bash
# Round-2 measurement pass on one MI355X box: everything profiles/r02_f_* is made from.
# usage (on the GPU box, repo root): bash tools/final_pass_r02.sh [tag=r02_f]
tag=${1:-r02_f}
O=gpurun_out
R=$PWD
set -x
timeout 900 python bench.py --save-profile $O/${tag}_decode_kernel_stats.txt > $O/${tag}_bench.json 2> $O/${tag}_bench.err
# the three --pmc passes behind roofline.traffic (separate passes, --kernel-trace only)
timeout 900 bash tools/pmc_pass.sh ${tag} > $O/${tag}_pmc.log 2>&1
# one 2048-token eval under the kernel trace (torch first: the profiler needs the runtime torch ships)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf1
LLAMAHIP_WITH_TORCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf1 -o p -- python $R/tools/prefill_one.py 2048 2 > /tmp/pf1.log 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/pf1 -name "*kernel_stats.csv") "rocprofv3 --kernel-trace --stats -- python tools/prefill_one.py 2048 2   (MI355X, synthetic LLaMA-7B Q4_0: model load, building the prompt copies, three 2048-token evals at n_ctx 2560; exact path, k_gemm_mfma16)" > $O/${tag}_prefill_2048_kernel_stats.txt
timeout 300 python tools/prefill_probe.py > $O/${tag}_prefill_probe.txt 2>&1
timeout 300 python tools/chunk_probe.py > $O/${tag}_chunk_probe.txt 2>&1
timeout 300 python tools/runner_probe.py > $O/${tag}_runner_probe.txt 2>&1
timeout 900 python bench.py --model 13B --no-insitu > $O/${tag}_bench_13B.json 2> $O/${tag}_bench_13B.err

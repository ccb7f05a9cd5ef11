set -x
python bench.py > gpurun_out/r01_i_bench.json 2> gpurun_out/r01_i_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o p1 -- python /root/repo/bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o p2 -- python /root/repo/tools/chunk9_probe.py > /dev/null 2>&1
cd /root/repo
python tools/prof_summary.py $(find /tmp/p1 -name "*kernel_stats.csv") "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline   (MI355X, synthetic LLaMA-7B Q4_0: hipGraph decode 6 launches per layer, the GEMV roofline launches, the prompt evaluations: one 504-token eval and the reference's 9-token chunks)" > gpurun_out/r01_i_decode_kernel_stats.txt
python tools/prof_summary.py $(find /tmp/p2 -name "*kernel_stats.csv") "rocprofv3 --kernel-trace --stats --output-format csv -- python tools/chunk9_probe.py   (MI355X, 7B: 56 evals of 9 tokens at growing n_past = the reference's prompt flow)" > gpurun_out/r01_i_chunk9_kernel_stats.txt
python tools/chunk_probe.py > gpurun_out/r01_i_chunk_probe.txt
python tools/prefill_probe.py > gpurun_out/r01_i_prefill_probe.txt 2>&1
python bench.py --model 13B > gpurun_out/r01_i_bench_13B.json 2> gpurun_out/r01_i_bench_13B.err

#!/bin/bash
# Round-4 pass D: the exact 2048-token prefill after templating the V*P kernel on "uniform split" (+ per-kernel stats), and the new
# full-width parity tests (13B / 65B widths across their schedule thresholds, full-depth 13B 128-token prompt)
O=gpurun_out; mkdir -p $O
python tools/prefill_one.py 2048 5 > $O/r04d_prefill_one.txt 2>&1; tail -1 $O/r04d_prefill_one.txt
rm -rf /tmp/pp; (cd /tmp && export TMPDIR=/tmp && LLAMAHIP_WITH_TORCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python /root/repo/tools/prefill_one.py 2048 2 > /dev/null 2>&1)
python tools/prof_summary.py $(find /tmp/pp -name "*kernel_stats.csv") "prefill_one.py 2048 2 (3 evals of 2048 tokens)" > $O/r04d_prefill_2048_kernel_stats.txt 2>&1; head -16 $O/r04d_prefill_2048_kernel_stats.txt
LLAMAHIP_SKIP_65B=1 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "wide_models or 13b_full_depth_128" --durations=5 > $O/r04d_fullsize_new.txt 2>&1; tail -8 $O/r04d_fullsize_new.txt

# Decode-attention A/B on the GPU box (measurement tooling): the decode parity subset, tools/decode_ab.sh over tools/ab_variants.txt,
# then the rocprofv3 kernel averages of the default build's decode step.  usage: bash tools/attn_ab.sh
timeout 300 python -m pytest tests -m gpu -x -q -k "trace_128 or graph or stage" 2>&1 | tail -3
timeout 300 tools/decode_ab.sh tools/ab_variants.txt 2>&1
cd /tmp && export TMPDIR=/tmp && LLAMAHIP_WITH_TORCH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa1 -o pa -- python /root/repo/tools/decode_probe.py --steps 64 --at 128 --reps 2 > /dev/null 2>&1; cd /root/repo; python tools/prof_summary.py $(find /tmp/pa1 -name "*kernel_stats.csv") | grep "k_dec\|k_gemv\|k_qkv"

# Where the fused wq|wk|wv + attention launch spends its time (measurement tooling): kernel averages with the polls of one or both
# hand-offs disabled (LLAMAHIP_ATTN_NOWAIT=1 all, 2 soft_max.V role only, 3 score role only -- RESULTS ARE INVALID in those runs, only the
# durations mean something).  usage: AT=128 bash tools/attn_nowait.sh
for v in "X=1" "LLAMAHIP_ATTN_NOWAIT=1" "LLAMAHIP_ATTN_NOWAIT=2" "LLAMAHIP_ATTN_NOWAIT=3"; do
echo "== $v"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa1 && env $v LLAMAHIP_WITH_TORCH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa1 -o pa -- python /root/repo/tools/decode_probe.py --steps 64 --at ${AT:-128} --reps 2 > /dev/null 2>&1); python tools/prof_summary.py $(find /tmp/pa1 -name "*kernel_stats.csv") | grep "k_qkv"
done

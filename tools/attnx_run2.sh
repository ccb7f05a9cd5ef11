for v in "LLAMAHIP_ATTN_NOWAIT=1" "LLAMAHIP_ATTN_NOWAIT=1 LLAMAHIP_NO_WO_FUSE=1"; do
echo "== $v"
env $v python tools/decode_probe.py --steps 64 --at 8,256 2>&1 | tail -1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa1 && env $v LLAMAHIP_WITH_TORCH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa1 -o pa -- python /root/repo/tools/decode_probe.py --steps 64 --at 128 --reps 2 > /dev/null 2>&1); python tools/prof_summary.py $(find /tmp/pa1 -name "*kernel_stats.csv") | grep "k_qkv\|k_gemv<0, 1, 16"
done

#!/bin/bash
# Round-3 first GPU pass: the overlapped two-branch decode schedule -- quick parity, A/B against the one-branch schedule, timelines.
O=gpurun_out; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "thread_splits or handoff_timeout" > $O/r03a_quick.txt 2>&1
tail -15 $O/r03a_quick.txt
cat > /tmp/v.txt <<EOV
base|LLAMAHIP_NO_OVERLAP=1
ov|LLAMAHIP_OVX=1
ov_resid22|LLAMAHIP_OV_DEPTH_RESID=22
ov_silu8|LLAMAHIP_OV_DEPTH_SILU=8
EOV
PROF=1 KEEP=1 STEPS=64 AT=8,256,440 PROF_AT=128 FILTER='k_gemv\|k_qkv\|k_embed\|k_argmax\|k_bump' timeout 1200 bash tools/decode_ab.sh /tmp/v.txt > $O/r03a_ab.txt 2>&1
cat $O/r03a_ab.txt
for v in base ov; do
  rm -rf /tmp/tl_$v
  envs=""; [ $v = base ] && envs="LLAMAHIP_NO_OVERLAP=1"
  (cd /tmp && export TMPDIR=/tmp && env LLAMAHIP_WITH_TORCH=1 $envs timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -o tl -- python $R/tools/decode_probe.py --steps 24 --at 128 --reps 1 > /tmp/tl_$v.log 2>&1)
  python tools/overlap_timeline.py /tmp/tl_$v --layers 3 > $O/r03a_timeline_$v.txt 2>&1
  cat $O/r03a_timeline_$v.txt
done
timeout 1200 python -m pytest tests -x -q -m gpu > $O/r03a_pytest.txt 2>&1
tail -8 $O/r03a_pytest.txt

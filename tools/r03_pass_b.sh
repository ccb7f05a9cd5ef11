#!/bin/bash
# Round-3 second GPU pass: where do the 35 us between consecutive launches of the second graph branch come from?
O=gpurun_out; mkdir -p $O; R=$PWD
cat > /tmp/v.txt <<EOV
base|LLAMAHIP_NO_OVERLAP=1
base_eager|LLAMAHIP_NO_OVERLAP=1 PROBE_FLAGS=1
ov|LLAMAHIP_OVX=1
ov_eager|PROBE_FLAGS=1
ov_nocapture|DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
ov_forceq|DEBUG_HIP_FORCE_GRAPH_QUEUES=4
ov_hwq8|GPU_MAX_HW_QUEUES=8
EOV
STEPS=64 AT=8,256 timeout 1200 bash tools/decode_ab.sh /tmp/v.txt > $O/r03b_ab.txt 2>&1
cat $O/r03b_ab.txt
for v in ov ov_eager; do
  rm -rf /tmp/tl_$v
  envs="LLAMAHIP_OVX=1"; [ $v = ov_eager ] && envs="PROBE_FLAGS=1"
  (cd /tmp && export TMPDIR=/tmp && env LLAMAHIP_WITH_TORCH=1 $envs timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -o tl -- python $R/tools/decode_probe.py --steps 24 --at 128 --reps 1 > /tmp/tl_$v.log 2>&1)
  python tools/overlap_timeline.py /tmp/tl_$v --layers 3 > $O/r03b_timeline_$v.txt 2>&1
  cat $O/r03b_timeline_$v.txt
done
timeout 600 python -m pytest tests/test_pipeline.py -x -q -m gpu > $O/r03b_pytest_pipeline.txt 2>&1
tail -5 $O/r03b_pytest_pipeline.txt

#!/bin/bash
# (as run at the time: LLAMAHIP_PV_STAGE / LLAMAHIP_ATTNQ_PF were tuning switches of that build; removed with the variants that lost)
# Round-3 GPU pass s: where k_dec_pv_stream's 16 us go (in-kernel timeline), prompt V*P prefetch depth
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fallback_paths and (switch1 or switch3)" > $O/r03s_quick.txt 2>&1; tail -3 $O/r03s_quick.txt
for sp in 1 2; do
  echo "== LLAMAHIP_PV_SPLIT=$sp"
  LLAMAHIP_PV_SPLIT=$sp timeout 600 python tools/pv_stream_timeline.py 2048 2 2>&1 | tail -32
done > $O/r03s_timeline.txt 2>&1
cat $O/r03s_timeline.txt
echo "== LLAMAHIP_PV_SPLIT=2 at 512" >> $O/r03s_timeline.txt
LLAMAHIP_PV_SPLIT=2 timeout 600 python tools/pv_stream_timeline.py 512 2 2>&1 | tail -32 >> $O/r03s_timeline.txt
tail -34 $O/r03s_timeline.txt
for pf in 4 8 12; do
  echo "== LLAMAHIP_ATTNQ_PF=$pf"
  LLAMAHIP_ATTNQ_PF=$pf timeout 600 python tools/prefill_one.py 2048 3 2>&1 | tail -1
done > $O/r03s_prefill_pf.txt 2>&1
cat $O/r03s_prefill_pf.txt

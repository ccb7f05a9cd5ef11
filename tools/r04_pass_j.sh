#!/bin/bash
# Round-4 pass J: k_dec_pv_dma (long-context soft_max . V with an LDS-DMA loader wave): parity + A/B against k_dec_pv_stream
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_everywhere or handoff_timeout or ragged_contexts" > $O/r04j_pytest.txt 2>&1; tail -5 $O/r04j_pytest.txt
LLAMAHIP_SKIP_65B=1 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "reference_flow or 13Bw_1600" >> $O/r04j_pytest.txt 2>&1; tail -4 $O/r04j_pytest.txt
cat > /tmp/variants.txt <<V
pv_stream|LLAMAHIP_PV_DMA=0
pv_dma|LLAMAHIP_X=1
V
PROF=1 PROF_AT=2048 N_CTX=2560 STEPS=48 AT=1408,2048,2400 FILTER='k_dec_pv\|k_dec_scores' tools/decode_ab.sh /tmp/variants.txt > $O/r04j_pv_dma_ab.txt 2>&1; cat $O/r04j_pv_dma_ab.txt

#!/bin/bash
# Round-4 pass A: first run of the persistent feed-forward launch (k_ffn_engine): parity subset, A/B against the three launches it
# replaces (tokens/s + per-kernel averages under rocprofv3), in-kernel timeline.
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny_model_golden or multipart or wider_models or thread_splits or dc_offset or 7b_logits or greedy_trace_128" > $O/r04a_pytest.txt 2>&1; tail -5 $O/r04a_pytest.txt
cat > /tmp/variants.txt <<V
launches|LLAMAHIP_NO_ENGINE=1
engine|LLAMAHIP_X=1
V
PROF=1 KEEP=1 STEPS=64 AT=8,256 FILTER='k_gemv\|k_qkv\|k_ffn\|k_embed\|k_argmax' tools/decode_ab.sh /tmp/variants.txt > $O/r04a_ffn_engine_ab.txt 2>&1; cat $O/r04a_ffn_engine_ab.txt
LLAMAHIP_LIB=libllamahip_probe3.so timeout 300 python tools/engine_timeline.py 128 3 > $O/r04a_engine_timeline.txt 2>&1; cat $O/r04a_engine_timeline.txt

#!/bin/bash
# Round-4 pass E: the lm head's pick epilogue (EPI_STORE_PICK: argmax + next embedding inside the output launch): parity + A/B
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny_model_golden or multipart or wider_models or thread_splits or 7b_logits or greedy_trace_128" > $O/r04e_pytest.txt 2>&1; tail -5 $O/r04e_pytest.txt
cat > /tmp/variants.txt <<V
two_launches|LLAMAHIP_NO_PICK_FOLD=1
pick_fold|LLAMAHIP_X=1
V
PROF=1 KEEP=1 STEPS=64 AT=8,256 FILTER='k_gemv<4, \|k_embed\|k_argmax' tools/decode_ab.sh /tmp/variants.txt > $O/r04e_pick_fold_ab.txt 2>&1; cat $O/r04e_pick_fold_ab.txt

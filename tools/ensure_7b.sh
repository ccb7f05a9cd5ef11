#!/bin/bash
# the synthetic 7B model file the probes read (same file the tests' model7b fixture makes)
D=${LLAMAHIP_MODEL_DIR:-/tmp/llamahip_models}/7B-seed20230312
if [ ! -f $D/ggml-model-q4_0.bin.done ]; then mkdir -p $D; llama.swift_amd/csrc/tools/make_synth_model --out $D/ggml-model-q4_0.bin --preset 7B --seed 20230312 > /dev/null && touch $D/ggml-model-q4_0.bin.done; fi

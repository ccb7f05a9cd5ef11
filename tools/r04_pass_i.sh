cat > /tmp/variants.txt <<V
base|LLAMAHIP_LIB=libllamahip_base.so
batched_polls|LLAMAHIP_LIB=libllamahip.so
V
PROF=1 STEPS=64 AT=8,256,440 FILTER='k_qkv' tools/decode_ab.sh /tmp/variants.txt 2>&1 | tee gpurun_out/r04i_batched_polls_ab.txt

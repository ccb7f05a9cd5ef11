#!/bin/bash
# Round-5 GPU passes (one parameterised script; the per-round r0N_pass_*.sh scripts of rounds 3-4 are gone).
# usage (on the GPU box, repo root):  bash tools/gpu_pass.sh <pass> [tag]
#   a      first pass of the few-row kernel (k_gemv_set): quick parity, A/B against k_gemm_skinny with per-kernel tables, plan sweep,
#          the nccl world-1 dry run of the pipeline bench
#   final  everything profiles/<tag>_* is made from (see the case below)
pass=${1:-a}; tag=${2:-r05_$pass}
O=gpurun_out; mkdir -p $O; R=$PWD
bash tools/ensure_7b.sh
case $pass in
a)
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -x -q -m gpu -k "mul_mat or tiny_model_golden or multipart or prompt_continuation or short_chunks or other_head or batched_set" --durations=5 > $O/${tag}_quick.txt 2>&1
  tail -25 $O/${tag}_quick.txt
  cat > /tmp/v.txt <<EOV
set|
skinny|LLAMAHIP_NO_GEMV_SET=1
EOV
  PROF=1 PROF_SEQS="4 8" PROF_EVALS=9 SEQS=2,4,8 EVALS=4,9,16 timeout 900 bash tools/set_ab.sh /tmp/v.txt > $O/${tag}_ab.txt 2>&1
  cat $O/${tag}_ab.txt
  cat > /tmp/v2.txt <<EOV
p41|LLAMAHIP_SET_PLAN=4,1
p22|LLAMAHIP_SET_PLAN=2,2
p14|LLAMAHIP_SET_PLAN=1,4
EOV
  PROF=1 PROF_SEQS="4" SEQS=4 EVALS= timeout 600 bash tools/set_ab.sh /tmp/v2.txt > $O/${tag}_plans4.txt 2>&1
  cat $O/${tag}_plans4.txt
  cat > /tmp/v3.txt <<EOV
p24|LLAMAHIP_SET_PLAN=2,4
p42|LLAMAHIP_SET_PLAN=4,2
p33|LLAMAHIP_SET_PLAN=3,3
p34|LLAMAHIP_SET_PLAN=3,4
EOV
  PROF=1 PROF_SEQS="8" SEQS=8 EVALS=9 timeout 700 bash tools/set_ab.sh /tmp/v3.txt > $O/${tag}_plans8.txt 2>&1
  cat $O/${tag}_plans8.txt
  # the RCCL branch of the pipeline bench at world 1: init_process_group("nccl", device_id), the three communicators, barrier, object
  # all-gather, the forced one-rank schedule in set mode
  LLAMAHIP_FORCE_PIPELINE=1 LLAMAHIP_BENCH_65B=0 LLAMAHIP_PIPE_NO_INSITU=1 timeout 600 python bench.py --steps 48 --warmup 4 > $O/${tag}_nccl_world1.json 2> $O/${tag}_nccl_world1.log
  grep -v "^$" $O/${tag}_nccl_world1.log | tail -25; head -c 1500 $O/${tag}_nccl_world1.json; echo
  ;;
b)
  # where the few-row kernel's time goes: ablation builds (make setab: no arithmetic | no weight loads in the loop | 8-deep ring)
  cat > /tmp/v.txt <<EOV
sa1|LLAMAHIP_LIB=libllamahip_sa1.so
sa2|LLAMAHIP_LIB=libllamahip_sa2.so
sa8|LLAMAHIP_LIB=libllamahip_sa8.so
sa1_p41|LLAMAHIP_LIB=libllamahip_sa1.so LLAMAHIP_SET_PLAN=4,1
sa2_p41|LLAMAHIP_LIB=libllamahip_sa2.so LLAMAHIP_SET_PLAN=4,1
sa8_p41|LLAMAHIP_LIB=libllamahip_sa8.so LLAMAHIP_SET_PLAN=4,1
EOV
  PROF=1 PROF_SEQS="4 8" PROF_EVALS=9 SEQS=4,8 EVALS=9 timeout 900 bash tools/set_ab.sh /tmp/v.txt > $O/${tag}_ablate.txt 2>&1
  cat $O/${tag}_ablate.txt
  ;;
c)
  # operand prefetch distance 3 (default build) against 1 (libllamahip_pf1.so), and the loop without its weight loads (sa2)
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -x -q -m gpu -k "mul_mat or batched_set or short_chunks" > $O/${tag}_quick.txt 2>&1; tail -3 $O/${tag}_quick.txt
  cat > /tmp/v.txt <<EOV
pf3|
pf1|LLAMAHIP_LIB=libllamahip_pf1.so
pf3_sa2|LLAMAHIP_LIB=libllamahip_sa2.so
pf3_p41|LLAMAHIP_SET_PLAN=4,1
pf3_p42|LLAMAHIP_SET_PLAN=4,2
EOV
  PROF=1 PROF_SEQS="4 8" PROF_EVALS=9 SEQS=4,8 EVALS=9 timeout 900 bash tools/set_ab.sh /tmp/v.txt > $O/${tag}_pf.txt 2>&1
  grep -v "k_repack\|copyBuffer\|k_argmax\|fillBuffer\|k_embed" $O/${tag}_pf.txt
  ;;
d)
  # generic A/B: variants from $VARIANTS (file), quick parity first
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -x -q -m gpu -k "mul_mat or batched_set or short_chunks" > $O/${tag}_quick.txt 2>&1; tail -3 $O/${tag}_quick.txt
  PROF=1 PROF_SEQS="${PROF_SEQS:-4 8}" PROF_EVALS=9 SEQS=${SEQS:-4,8} EVALS=9 timeout 900 bash tools/set_ab.sh ${VARIANTS:-tools/variants.txt} > $O/${tag}_ab.txt 2>&1
  grep -v "k_repack\|copyBuffer\|k_argmax\|fillBuffer\|k_embed" $O/${tag}_ab.txt
  if [ -n "$TIMELINE" ]; then bash tools/gpu_pass.sh t ${tag}; fi
  ;;
m)
  # mid-round pass: the new GPU tests, single-stream decode experiments (k_qkv_attn timeline of the tree, q|k-first
  # dispatch order A/B), kernel tables + PMC traffic of the set step, the reference's 9-token evals
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -x -q -m gpu -k "mul_mat or batched_set or short_chunks or few_row_handoff or handoff_timeout or prompt_continuation or chunks_in_one_pass" > $O/${tag}_quick.txt 2>&1; tail -3 $O/${tag}_quick.txt
  cat > /tmp/v.txt <<EOV
base|
qk_first|LLAMAHIP_QKV_QK_FIRST=1
EOV
  PROF=1 KEEP=1 STEPS=96 AT=8,256,440 PROF_AT=128 FILTER='k_gemv\|k_qkv\|k_embed' timeout 600 bash tools/decode_ab.sh /tmp/v.txt > $O/${tag}_qkv_order_ab.txt 2>&1
  cat $O/${tag}_qkv_order_ab.txt
  timeout 300 python tools/attn_timeline.py 128 3 > $O/${tag}_attn_timeline.txt 2>&1; tail -30 $O/${tag}_attn_timeline.txt
  cat > /tmp/v2.txt <<EOV
set|
skinny|LLAMAHIP_NO_GEMV_SET=1
EOV
  PROF=1 PROF_SEQS="4 8" PROF_EVALS=9 SEQS=2,4,8 EVALS=4,9,16 timeout 900 bash tools/set_ab.sh /tmp/v2.txt > $O/${tag}_set_ab.txt 2>&1
  grep -v "k_repack\|copyBuffer\|fillBuffer" $O/${tag}_set_ab.txt
  timeout 400 bash tools/pmc_set_pass.sh 4 > $O/${tag}_set_pmc_S4.txt 2>&1; cat $O/${tag}_set_pmc_S4.txt
  timeout 400 bash tools/pmc_set_pass.sh 8 > $O/${tag}_set_pmc_S8.txt 2>&1; cat $O/${tag}_set_pmc_S8.txt
  ;;
t)
  # in-kernel timelines of the few-row kernel (libllamahip_setprobe.so)
  for spec in "--seqs 4" "--seqs 8" "--evals 9"; do
    for plan in ${PLANS:-""}; do
      echo "== set_timeline $spec plan=[$plan]"
      LLAMAHIP_SET_PLAN=$plan timeout 300 python tools/set_timeline.py $spec 2>&1 | tail -12
    done
  done > $O/${tag}_timeline.txt 2>&1
  cat $O/${tag}_timeline.txt
  ;;
esac

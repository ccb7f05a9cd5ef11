#!/bin/bash
# The GPU passes of a round in ONE parameterised script (rounds 3-4 kept thirty one-shot r0N_pass_*.sh files; they are gone).
# usage (on the GPU box, repo root):  bash tools/gpu_pass.sh <pass> [tag]      -- outputs under gpurun_out/<tag>_*
#   quick  parity subset of the few-row paths (mat-mul shapes, set steps, short evals, fault injection)
#   ab     A/B of library / switch variants: VARIANTS=<file of "label|ENV=VAL ..." lines> (tools/set_ab.sh: per-kernel tables with PROF=1,
#          tools/fresh_ab.sh: every measurement in a fresh process), TIMELINE=1 adds the in-kernel timelines
#   t      in-kernel timelines of k_gemv_set (libllamahip_setprobe.so: tools/build_set_variants.sh setprobe:"-DLH_SET_PROBE=1")
#   mid    single-stream decode experiments (k_qkv_attn timeline, decode A/B over VARIANTS), set-step kernel tables + PMC traffic
#   x      soft_max ablation of k_dec_pv_dma at 2 048 keys (libllamahip_pvabl.so)
#   pf     long-prompt A/B of library / switch variants (VARIANTS) + the prompt parity tests
#   w13    whole- vs half-block w1|w3 workgroups of the few-row kernel + 9-row plans (fresh processes), the variant parity test
#   verify the last check of a tree: full GPU suite, smoke(), bench.py with defaults, chunk / prefill probes
#   nccl   the RCCL branch of the pipeline bench at world 1 (communicators, self-check, forced one-rank schedule) with its log
#   65b    BASELINE configs[4]'s model on one GPU: the forced one-rank pipeline in set mode (in-situ roofline of the stage step, parity gate)
#   final  everything profiles/<tag>_* is made from: full GPU test suite, bench.py (7B, 13B), config[3] mixed run with HBM counters,
#          prefill kernel table, prompt / chunk / runner probes, set-step tables + PMC, fresh-process set / eval rates
pass=${1:-quick}; tag=${2:-r06_$pass}
O=gpurun_out; mkdir -p $O; R=$PWD
bash tools/ensure_7b.sh
quick() { timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -x -q -m gpu -k "mul_mat or batched_set or short_chunks or few_row or handoff_timeout or prompt_continuation or chunks_in_one_pass" --durations=5 > $O/${tag}_quick.txt 2>&1; tail -4 $O/${tag}_quick.txt; }
set_tables() {      # kernel tables of the set step (4 and 8 sequences, then 9-token evals) + PMC traffic (k_gemm_skinny, the baseline of round 5's A/Bs, is gone)
  printf 'set|\n' > /tmp/v_set.txt
  PROF=1 PROF_SEQS="4 8" PROF_EVALS=9 SEQS= EVALS= timeout 900 bash tools/set_ab.sh /tmp/v_set.txt > $O/${tag}_set_step_kernel_stats.txt 2>&1
  grep -v "k_repack\|copyBuffer\|fillBuffer" $O/${tag}_set_step_kernel_stats.txt | head -80
  SEQS="2 4 8" EVALS="4 9 16" timeout 600 bash tools/fresh_ab.sh /tmp/v_set.txt > $O/${tag}_set_fresh_process_ab.txt 2>&1; cat $O/${tag}_set_fresh_process_ab.txt
  for S in 4 8; do timeout 400 bash tools/pmc_set_pass.sh $S > $O/${tag}_set_pmc_S$S.txt 2>&1; head -12 $O/${tag}_set_pmc_S$S.txt; done
}
case $pass in
quick) quick ;;
ab)
  quick
  PROF=${PROF-1} PROF_SEQS="${PROF_SEQS:-4 8}" PROF_EVALS=9 SEQS=${SEQS:-4,8} EVALS=9 timeout 900 bash tools/set_ab.sh ${VARIANTS:-tools/variants.txt} > $O/${tag}_ab.txt 2>&1
  grep -v "k_repack\|copyBuffer\|k_argmax\|fillBuffer\|k_embed" $O/${tag}_ab.txt
  timeout 600 bash tools/fresh_ab.sh ${VARIANTS:-tools/variants.txt} > $O/${tag}_fresh_ab.txt 2>&1; cat $O/${tag}_fresh_ab.txt
  if [ -n "$TIMELINE" ]; then bash tools/gpu_pass.sh t ${tag}; fi
  ;;
t)
  for spec in "--seqs 4" "--seqs 8" "--evals 9"; do
    for plan in ${PLANS:-""}; do
      echo "== set_timeline $spec plan=[$plan]"
      LLAMAHIP_SET_PLAN=$plan timeout 300 python tools/set_timeline.py $spec 2>&1 | tail -12
    done
  done > $O/${tag}_timeline.txt 2>&1
  cat $O/${tag}_timeline.txt
  ;;
mid)
  quick
  PROF=1 KEEP=1 STEPS=96 AT=8,256,440 PROF_AT=128 FILTER='k_gemv\|k_qkv\|k_embed' timeout 600 bash tools/decode_ab.sh ${VARIANTS:-tools/variants.txt} > $O/${tag}_decode_ab.txt 2>&1
  cat $O/${tag}_decode_ab.txt
  timeout 300 python tools/attn_timeline.py 128 3 > $O/${tag}_attn_timeline.txt 2>&1; tail -8 $O/${tag}_attn_timeline.txt
  set_tables
  ;;
nccl)
  LLAMAHIP_FORCE_PIPELINE=1 LLAMAHIP_BENCH_65B=0 timeout 900 python bench.py --steps 64 --warmup 8 > $O/${tag}_nccl_world1.json 2> $O/${tag}_nccl_world1.log
  grep -v "^$" $O/${tag}_nccl_world1.log | tail -25; head -c 2500 $O/${tag}_nccl_world1.json; echo
  ;;
x)    # round 5's last bounded experiment still in the tree: the soft_max of k_dec_pv_dma reduced to a workgroup's own keys (upper bound of a split;
      # SRC=decode tools/build_set_variants.sh pvabl:"-DLH_PVD_ABLATE=1").  (The quarter-block w1|w3 A/B of the same pass: profiles/r05_q_w13_quarter.diff.)
  printf 'product|\nsoft_max_own_keys_only|LLAMAHIP_LIB=libllamahip_pvabl.so\n' > /tmp/v_p.txt
  PROF=1 STEPS=48 AT=2048 PROF_AT=2048 N_CTX=2560 FILTER='k_dec_pv\|k_dec_scores' timeout 600 bash tools/decode_ab.sh /tmp/v_p.txt > $O/${tag}_pv_softmax_ablation.txt 2>&1; cat $O/${tag}_pv_softmax_ablation.txt
  ;;
pf)   # long-prompt A/B over VARIANTS (tools/prefill_ab.sh) + the prompt parity tests on the product build
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "prompt_continuation or chunks_in_one_pass or 2048_token_prefill or reference_flow" --durations=5 > $O/${tag}_prompt_parity.txt 2>&1; tail -4 $O/${tag}_prompt_parity.txt
  PROF=1 N=2048 timeout 900 bash tools/prefill_ab.sh ${VARIANTS:-tools/variants.txt} > $O/${tag}_prefill_ab.txt 2>&1; cat $O/${tag}_prefill_ab.txt
  ;;
w13)  # whole-block vs half-block w1|w3 workgroups and the 9-row plans, fresh processes; the variant parity test
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "few_row_kernel_selectable or few_row_handoff or mul_mat" > $O/${tag}_parity.txt 2>&1; tail -3 $O/${tag}_parity.txt
  SEQS="${SEQS:-4 8}" EVALS="${EVALS:-9 16}" timeout 900 bash tools/fresh_ab.sh ${VARIANTS:-tools/variants.txt} > $O/${tag}_fresh_ab.txt 2>&1; cat $O/${tag}_fresh_ab.txt
  ;;
verify)   # the last check of a tree: full GPU suite, smoke(), bench.py with defaults, the chunk / prefill probes
  python -m pytest tests -x -q -m gpu --durations=8 > $O/${tag}_pytest.txt 2>&1; tail -12 $O/${tag}_pytest.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout 1200 python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.log; tail -2 $O/${tag}_bench.log; python - <<PY
import json
d = json.load(open("$O/${tag}_bench.json"))
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step")}, d["roofline"].get("frac"), d["roofline"].get("end_to_end_frac"), d["roofline"].get("traffic"))
print("config", {k: v for k, v in d["config"].items() if not isinstance(v, (dict, list))})
print("parity", d.get("parity", {}).get("identical"), d.get("parity", {}).get("tokens_compared"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
  timeout 300 python tools/chunk_probe.py > $O/${tag}_chunk_probe.txt 2>&1; tail -7 $O/${tag}_chunk_probe.txt
  timeout 300 python tools/prefill_probe.py > $O/${tag}_prefill_probe.txt 2>&1; tail -6 $O/${tag}_prefill_probe.txt
  ;;
65b)
  LLAMAHIP_FORCE_PIPELINE=1 LLAMAHIP_PIPE_PARITY_S=${PARITY_S:-60} timeout 2400 python bench.py --model 65B --steps 32 --warmup 4 > $O/${tag}_bench_65B_1gpu.json 2> $O/${tag}_bench_65B_1gpu.log
  tail -12 $O/${tag}_bench_65B_1gpu.log; head -c 3000 $O/${tag}_bench_65B_1gpu.json; echo
  ;;
final)
  # (LLAMAHIP_HEAD: the commit the snapshot was taken from -- there is no .git on the GPU box; the caller passes `git rev-parse HEAD`)
  { echo "# tree: ${LLAMAHIP_HEAD:-unknown} -- python -m pytest tests -x -q -m gpu --durations=12 on $(date -u +%Y-%m-%dT%H:%MZ)"; python -m pytest tests -x -q -m gpu --durations=12; } > $O/${tag}_pytest.txt 2>&1; tail -16 $O/${tag}_pytest.txt
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout 1200 python bench.py --save-profile $O/${tag}_decode_kernel_stats.txt > $O/${tag}_bench.json 2> $O/${tag}_bench.log; tail -3 $O/${tag}_bench.log; python - <<PY
import json
d = json.load(open("$O/${tag}_bench.json"))
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step")}, d["roofline"].get("frac"), d["roofline"].get("end_to_end_frac"), d["roofline"].get("traffic"))
print("config", d["config"])
print("roofline scalars", {k: v for k, v in d["roofline"].items() if k.startswith("dominant_by_time_") or k.startswith("full_context")})
print("full_context", d.get("full_context", {}).get("tokens_per_s"), "prefill2048", d["prefill"].get("configs2_2048_tokens_one_eval", {}).get("tokens_per_s"),
      "decode after", d["prefill"].get("configs2_2048_tokens_one_eval", {}).get("decode_after_prompt"))
print("concurrent", [c.get("aggregate_tokens_per_s") for c in d.get("concurrent_sequences", [])] if isinstance(d.get("concurrent_sequences"), list) else d.get("concurrent_sequences"))
print("batched", [(c.get("sequences"), c.get("aggregate_tokens_per_s"), c.get("tokens_equal_single_stream")) for c in d.get("batched_sequences", [])] if isinstance(d.get("batched_sequences"), list) else d.get("batched_sequences"))
print("parity", d.get("parity"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
  # configs[3]: 13B bench line (parity gate + cpu_baseline) and the mixed prefill + decode run under rocprofv3 (decode crosses position 544)
  timeout 1200 python bench.py --model 13B --no-prefill-2048 --no-concurrent > $O/${tag}_bench_13B.json 2> $O/${tag}_bench_13B.log; python - <<PY
import json
d = json.load(open("$O/${tag}_bench_13B.json"))
print("13B", {k: d[k] for k in ("value", "ms_per_step")}, d["roofline"].get("frac"), d["roofline"].get("end_to_end_frac"), d.get("parity"), d.get("cpu_baseline", {}).get("value"), d["config"])
PY
  timeout 1200 bash tools/run_config4.sh 13B > $O/${tag}_cfg4.log 2>&1; { grep -h "prefill\|decode" $O/cfg4/run_plain.txt | sed 's/^/# /'; cat $O/cfg4/hbm_summary.txt; } > $O/${tag}_13B_mixed_hbm.txt; head -24 $O/${tag}_13B_mixed_hbm.txt
  # prompt paths
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pf1
  LLAMAHIP_WITH_TORCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf1 -o p -- python $R/tools/prefill_one.py 2048 2 > /tmp/pf1.log 2>&1
  cd $R
  python tools/prof_summary.py $(find /tmp/pf1 -name "*kernel_stats.csv") "rocprofv3 --kernel-trace --stats -- python tools/prefill_one.py 2048 2   (MI355X, synthetic LLaMA-7B Q4_0: model load, building the prompt copies, three 2048-token evals at n_ctx 2560; exact path, k_gemm_mfma4)" > $O/${tag}_prefill_2048_kernel_stats.txt
  timeout 300 python tools/prefill_probe.py > $O/${tag}_prefill_probe.txt 2>&1; tail -6 $O/${tag}_prefill_probe.txt
  timeout 300 python tools/chunk_probe.py > $O/${tag}_chunk_probe.txt 2>&1; tail -7 $O/${tag}_chunk_probe.txt
  timeout 300 python tools/runner_probe.py > $O/${tag}_runner_probe.txt 2>&1; tail -3 $O/${tag}_runner_probe.txt
  set_tables
  timeout 300 python tools/attn_timeline.py 128 3 > $O/${tag}_attn_timeline.txt 2>&1; tail -6 $O/${tag}_attn_timeline.txt
  # the in-process layer pipeline (one llamahip_model_load with a device list) on this one GPU, and every launch of the default schedules with its scratch
  timeout 600 python tools/inprocess_probe.py 7B 1,2,4,8 128 > $O/${tag}_inprocess_pipeline_7B.txt 2>&1; cat $O/${tag}_inprocess_pipeline_7B.txt
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sw
  LLAMAHIP_WITH_TORCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sw -o w -- python $R/tools/schedule_walk.py > /tmp/sw.log 2>&1; cd $R
  cp $(find /tmp/sw -name "*kernel_stats.csv" | head -1) $O/${tag}_schedule_walk_kernel_stats.csv 2>/dev/null; tail -2 /tmp/sw.log | cut -c1-160
  ;;
esac

#!/bin/bash
# Round-4 final pass on one MI355X box: everything profiles/r04_final_* is made from.
# usage (on the GPU box, repo root): bash tools/final_pass_r04.sh [tag=r04_final]
tag=${1:-r04_final}
O=gpurun_out; mkdir -p $O
R=$PWD
python -m pytest tests -x -q -m gpu --durations=8 > $O/${tag}_pytest.txt 2>&1; tail -12 $O/${tag}_pytest.txt
timeout 1200 python bench.py --save-profile $O/${tag}_decode_kernel_stats.txt > $O/${tag}_bench.json 2> $O/${tag}_bench.log; tail -3 $O/${tag}_bench.log; python - <<PY
import json
d = json.load(open("$O/${tag}_bench.json"))
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step")}, d["roofline"].get("frac"), d["roofline"].get("end_to_end_frac"), d["roofline"].get("traffic"))
print("config", d["config"])
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
print("13B", {k: d[k] for k in ("value", "ms_per_step")}, d["roofline"].get("frac"), d["roofline"].get("end_to_end_frac"), d.get("parity"), d.get("cpu_baseline", {}).get("value"))
PY
timeout 1200 bash tools/run_config4.sh 13B > $O/${tag}_cfg4.log 2>&1; { grep -h "prefill\|decode" $O/cfg4/run_plain.txt | sed 's/^/# /'; cat $O/cfg4/hbm_summary.txt; } > $O/${tag}_13B_mixed_hbm.txt; head -24 $O/${tag}_13B_mixed_hbm.txt
# prompt paths
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pf1
LLAMAHIP_WITH_TORCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf1 -o p -- python $R/tools/prefill_one.py 2048 2 > /tmp/pf1.log 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/pf1 -name "*kernel_stats.csv") "rocprofv3 --kernel-trace --stats -- python tools/prefill_one.py 2048 2   (MI355X, synthetic LLaMA-7B Q4_0: model load, building the prompt copies, three 2048-token evals at n_ctx 2560; exact path, k_gemm_mfma4)" > $O/${tag}_prefill_2048_kernel_stats.txt
timeout 300 python tools/prefill_probe.py > $O/${tag}_prefill_probe.txt 2>&1; tail -6 $O/${tag}_prefill_probe.txt
timeout 300 python tools/chunk_probe.py > $O/${tag}_chunk_probe.txt 2>&1; tail -4 $O/${tag}_chunk_probe.txt
timeout 300 python tools/runner_probe.py > $O/${tag}_runner_probe.txt 2>&1; tail -3 $O/${tag}_runner_probe.txt

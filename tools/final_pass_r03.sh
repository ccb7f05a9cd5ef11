#!/bin/bash
# Round-3 final pass: the whole GPU test suite + the default bench (with its profile) on the final build
TAG=${1:-r03y}; O=gpurun_out; mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=8 > $O/${TAG}_pytest.txt 2>&1; tail -5 $O/${TAG}_pytest.txt
python bench.py --save-profile $O/${TAG}_decode_kernel_stats.txt > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; tail -3 $O/${TAG}_bench.log; python - <<PY
import json
d = json.load(open("$O/${TAG}_bench.json"))
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step")}, d["roofline"].get("frac"), d["roofline"].get("end_to_end_frac"))
print("full_context", d.get("full_context", {}).get("tokens_per_s"), "prefill2048", d["prefill"].get("configs2_2048_tokens_one_eval", {}).get("tokens_per_s"),
      "decode after", d["prefill"].get("configs2_2048_tokens_one_eval", {}).get("decode_after_prompt"))
print("concurrent", [c.get("tokens_per_s") for c in d.get("concurrent_sequences", [])] if isinstance(d.get("concurrent_sequences"), list) else d.get("concurrent_sequences"))
print("parity", d.get("parity"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY

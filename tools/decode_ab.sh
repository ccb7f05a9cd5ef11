#!/bin/bash
# A/B of decode variants on the GPU box (measurement tooling).  Each line of the here-doc / file given as $1 is
# "label|ENV=VAL ENV2=VAL ..."; every variant runs tools/decode_probe.py in its own process (the switches are read
# once per process) and, with PROF=1, once more under rocprofv3 for the per-kernel averages of its decode step.
# usage: PROF=1 STEPS=64 AT=8,256 [N_CTX=512 MODEL=7B] tools/decode_ab.sh variants.txt
cd "$(dirname "$0")/.."
STEPS=${STEPS:-64}; AT=${AT:-8,256}; N_CTX=${N_CTX:-512}; MODEL=${MODEL:-7B}; FILTER=${FILTER:-k_gemv\|k_dec_\|k_embed\|k_argmax}
python tools/decode_probe.py --model $MODEL --steps 4 --at 8 --reps 1 > /dev/null 2>&1      # writes the model file once
while IFS='|' read -r label envs; do
  [ -z "$label" ] && continue
  echo "== $label   [$envs]"
  env $envs python tools/decode_probe.py --model $MODEL --n_ctx $N_CTX --steps $STEPS --at $AT 2>&1 | tail -1
  if [ -n "$PROF" ]; then
    rm -rf /tmp/pa_$$
    (cd /tmp && export TMPDIR=/tmp && env LLAMAHIP_WITH_TORCH=1 $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_$$ -o pa -- python /root/repo/tools/decode_probe.py --model $MODEL --n_ctx $N_CTX --steps $STEPS --at ${PROF_AT:-128} --reps 2 > /dev/null 2>&1)
    python tools/prof_summary.py $(find /tmp/pa_$$ -name "*kernel_stats.csv") | grep "$FILTER"
    [ -n "$KEEP" ] && mkdir -p gpurun_out/ab && python tools/prof_summary.py $(find /tmp/pa_$$ -name "*kernel_stats.csv") "$label [$envs] decode_probe --steps $STEPS --at ${PROF_AT:-128}" > gpurun_out/ab/$label.txt
    rm -rf /tmp/pa_$$
  fi
done < "${1:-/dev/stdin}"

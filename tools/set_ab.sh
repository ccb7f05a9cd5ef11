#!/bin/bash
# A/B of the few-row paths on the GPU box (measurement tooling): each line of the file given as $1 (or stdin) is "label|ENV=VAL ...";
# every variant runs tools/set_probe.py in its own process and, with PROF=1, once more under rocprofv3 for the per-kernel table of
# its set steps (written to gpurun_out/set_ab/<label>.txt).
# usage: PROF=1 SEQS=4,8 EVALS=9 STEPS=96 tools/set_ab.sh variants.txt
cd "$(dirname "$0")/.."
SEQS=${SEQS:-4,8}; EVALS=${EVALS:-9}; STEPS=${STEPS:-96}; MODEL=${MODEL:-7B}; N_CTX=${N_CTX:-512}
mkdir -p gpurun_out/set_ab
while IFS='|' read -r label envs; do
  [ -z "$label" ] && continue
  echo "== $label   [$envs]"
  env $envs python tools/set_probe.py --model $MODEL --seqs $SEQS --steps $STEPS --n_ctx $N_CTX --evals "$EVALS" 2>&1 | grep "set of\|evals of\|Error\|error" | sed 's/, paths.*//'
  if [ -n "$PROF" ]; then
    for S in ${PROF_SEQS:-4}; do
      rm -rf /tmp/sa_$$
      (cd /tmp && export TMPDIR=/tmp && env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sa_$$ -o sa -- python /root/repo/tools/set_probe.py --model $MODEL --seqs $S --steps $STEPS --n_ctx $N_CTX ${PROF_EVALS:+--evals $PROF_EVALS} > /dev/null 2>&1)
      f=$(find /tmp/sa_$$ -name "*kernel_stats.csv" | head -1)
      if [ -n "$f" ]; then
        python tools/prof_summary.py $f "$label [$envs] rocprofv3 --kernel-trace --stats -- tools/set_probe.py --model $MODEL --seqs $S --steps $STEPS ${PROF_EVALS:+--evals $PROF_EVALS}" > gpurun_out/set_ab/${label}_S$S.txt
        head -14 gpurun_out/set_ab/${label}_S$S.txt
      else echo "(no kernel stats for $label S=$S)"; fi
      rm -rf /tmp/sa_$$
    done
  fi
done < "${1:-/dev/stdin}"

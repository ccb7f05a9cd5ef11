#!/bin/bash
# set steps and short evals, every measurement in a FRESH process (a handle loaded after another one was closed in the same process decodes
# ~20 % slower here -- device memory comes back fragmented), for each "label|ENV=VAL ..." line of $1 (e.g. "product|" and library / plan variants)
cd "$(dirname "$0")/.."
V=${1:-/dev/stdin}
[ -f "$V" ] || V=/dev/stdin
while IFS='|' read -r label envs; do
  [ -z "$label" ] && continue
  echo "== $label   [$envs]"
  for S in ${SEQS:-2 4 8}; do env $envs python tools/set_probe.py --seqs $S 2>&1 | grep "set of" | sed "s/, paths.*//"; done
  for n in ${EVALS:-4 9 16}; do env $envs python tools/set_probe.py --seqs "" --evals $n 2>&1 | grep evals; done
done < "$V"

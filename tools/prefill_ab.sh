#!/bin/bash
# A/B of long-prompt variants on the GPU box (measurement tooling).  Each line of $1 is "label|ENV=VAL ..." (e.g. LLAMAHIP_LIB=libllamahip_x.so);
# every variant runs tools/prefill_one.py N 3 in its own process and, with PROF=1, once more under rocprofv3 for the per-kernel table.
# usage: PROF=1 N=2048 FILTER='k_attnq' tools/prefill_ab.sh variants.txt
cd "$(dirname "$0")/.."
N=${N:-2048}; FILTER=${FILTER:-k_attnq\|k_gemm\|k_prep}
while IFS='|' read -r label envs; do
  [ -z "$label" ] && continue
  echo "== $label   [$envs]"
  env $envs python tools/prefill_one.py $N 3 2>&1 | tail -1
  if [ -n "$PROF" ]; then
    rm -rf /tmp/pf_$$
    (cd /tmp && export TMPDIR=/tmp && env LLAMAHIP_WITH_TORCH=1 $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$$ -o p -- python /root/repo/tools/prefill_one.py $N 2 > /dev/null 2>&1)
    python tools/prof_summary.py $(find /tmp/pf_$$ -name "*kernel_stats.csv") | grep "calls\|$FILTER"
    rm -rf /tmp/pf_$$
  fi
done < "${1:-/dev/stdin}"

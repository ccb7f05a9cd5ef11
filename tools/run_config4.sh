#!/bin/bash
# BASELINE.json configs[3]: LLaMA-13B mixed prefill+decode with rocprofv3 HBM capture (run on the GPU box via gpurun)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/cfg4; mkdir -p $OUT
MODEL=${1:-13B}
python $R/tools/mixed_run.py $MODEL > $OUT/run_plain.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -o a -- python $R/tools/mixed_run.py $MODEL > $OUT/run_a.txt 2>&1
LLAMAHIP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/b -o b -- python $R/tools/mixed_run.py $MODEL > $OUT/run_b.txt 2>&1
LLAMAHIP_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c -o c -- python $R/tools/mixed_run.py $MODEL > $OUT/run_c.txt 2>&1
cd $R
python tools/hbm_summary.py $(find $OUT/a -name '*kernel_stats.csv' | head -1) $(find $OUT/b -name '*counter_collection.csv' | head -1) $(find $OUT/c -name '*counter_collection.csv' | head -1) "LLaMA-$MODEL Q4_0 mixed prefill+decode (tools/mixed_run.py), MI355X, rocprofv3" > $OUT/hbm_summary.txt 2>&1
cat $OUT/run_plain.txt; cat $OUT/hbm_summary.txt
# keep the merge small: the raw traces are large
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*agent_info.csv' -delete

#!/bin/bash
# Round-4 pass R: 9-token chunks: kernel breakdown + column-group width (waves on the chip) A/B
O=gpurun_out; mkdir -p $O
bash tools/ensure_7b.sh
{
for wv in 1536 3072 6144; do echo "== LLAMAHIP_SKINNY_WAVES=$wv"; LLAMAHIP_SKINNY_WAVES=$wv timeout 300 python tools/chunk_probe.py 4 8 9 16 32 2>&1 | grep -v amdgpu.ids; done
} > $O/r04r_skinny_waves_ab.txt 2>&1; cat $O/r04r_skinny_waves_ab.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pc9
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc9 -o p -- python $R/tools/chunk9_probe.py > /tmp/pc9.log 2>&1
cd $R; tail -2 /tmp/pc9.log
python tools/prof_summary.py $(find /tmp/pc9 -name "*kernel_stats.csv") "rocprofv3 --kernel-trace --stats -- python tools/chunk9_probe.py   (56 evals of 9 tokens, 7B)" > $O/r04r_chunk9_kernel_stats.txt; head -24 $O/r04r_chunk9_kernel_stats.txt

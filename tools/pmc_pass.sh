#!/bin/bash
# The three rocprofv3 --pmc passes behind roofline.traffic (separate passes, --kernel-trace only), summarised
# into gpurun_out/<tag>_gemv_pmc.txt and gpurun_out/gemv_traffic.json.  usage: tools/pmc_pass.sh <tag>
tag=${1:-r01}
python bench.py --steps 8 --no-cpu-baseline >/dev/null 2>&1      # makes sure the synthetic 7B file exists
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1 /tmp/pmc2 /tmp/pmc3
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -o p -- python $R/tools/gemv_probe.py 4 > /tmp/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc2 -o p -- python $R/tools/gemv_probe.py 4 > /tmp/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc3 -o p -- python $R/tools/gemv_probe.py 4 > /tmp/pmc3.log 2>&1
cd $R
python tools/pmc_summary.py $(find /tmp/pmc1 -name "*counter_collection.csv") $(find /tmp/pmc2 -name "*counter_collection.csv") $(find /tmp/pmc3 -name "*counter_collection.csv") gpurun_out/${tag}_gemv_pmc.txt gpurun_out/gemv_traffic.json
cat gpurun_out/${tag}_gemv_pmc.txt

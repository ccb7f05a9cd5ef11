#!/bin/bash
# Round-4 pass T: counters of the 2048-token prefill's attention kernels (two separate --pmc passes, --kernel-trace only)
O=gpurun_out; mkdir -p $O
bash tools/ensure_7b.sh
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pm1 /tmp/pm2
LLAMAHIP_WITH_TORCH=1 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pm1 -o p -- python $R/tools/prefill_one.py 2048 1 > /tmp/pm1.log 2>&1
LLAMAHIP_WITH_TORCH=1 timeout 400 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pm2 -o p -- python $R/tools/prefill_one.py 2048 1 > /tmp/pm2.log 2>&1
cd $R; tail -3 /tmp/pm1.log /tmp/pm2.log
{ echo "# rocprofv3 --pmc (two passes) --kernel-trace -- python tools/prefill_one.py 2048 1: averages per launch, summed over the chip"; python tools/pmc_generic.py $(find /tmp/pm1 -name "*counter_collection.csv"); python tools/pmc_generic.py $(find /tmp/pm2 -name "*counter_collection.csv"); } > $O/r04t_prefill_pmc.txt 2>&1; cat $O/r04t_prefill_pmc.txt

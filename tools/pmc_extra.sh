#!/bin/bash
# Extra PMC passes over the bench (counters only, no trace domain): where a kernel WAITS -- scalar-cache misses, LDS issue stalls, memory instruction counts.
#   bash tools/pmc_extra.sh r03c      ->  gpurun_out/pmcx_<tag>_{a,b,c}/   (summarise: python tools/rocpd_pmc_summary.py gpurun_out/pmcx_<tag>_*/pmc_results.db)
set -u
tag=$1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY -d $R/gpurun_out/pmcx_${tag}_a -o pmc -- python $R/bench.py --no-cpu-baseline --no-steady-state --steps 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $R/gpurun_out/pmcx_${tag}_b -o pmc -- python $R/bench.py --no-cpu-baseline --no-steady-state --steps 5 > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM -d $R/gpurun_out/pmcx_${tag}_c -o pmc -- python $R/bench.py --no-cpu-baseline --no-steady-state --steps 5 > /dev/null 2>&1
cd $R
python tools/rocpd_pmc_summary.py gpurun_out/pmcx_${tag}_a/pmc_results.db gpurun_out/pmcx_${tag}_b/pmc_results.db gpurun_out/pmcx_${tag}_c/pmc_results.db > gpurun_out/pmcx_${tag}.md 2>&1
cat gpurun_out/pmcx_${tag}.md

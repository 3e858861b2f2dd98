#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repository root):
#   bash tools/profile_round.sh r01f
# writes gpurun_out/prof_<tag>/ (kernel trace of `bench.py --no-cpu-baseline`), gpurun_out/pmc_<tag>_{sq,fetch,write}/ (PMC passes,
# counters only -- never combined with a trace domain) and gpurun_out/bench_<tag>.json (the full bench line, CPU baseline included).
# Afterwards, here: python tools/make_profile_summaries.py <tag>   ->  profiles/<tag>_*.
set -u
tag=$1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o bench -- python $R/bench.py --no-cpu-baseline --no-steady-state --no-overlap > $R/gpurun_out/prof_$tag.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_${tag}_sq -o pmc -- python $R/bench.py --no-cpu-baseline --no-steady-state --no-overlap --steps 5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${tag}_fetch -o pmc -- python $R/bench.py --no-cpu-baseline --no-steady-state --no-overlap --steps 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${tag}_write -o pmc -- python $R/bench.py --no-cpu-baseline --no-steady-state --no-overlap --steps 5 > /dev/null 2>&1
# the default command (WBC of step k next to the node kernels of step k + 1): one more kernel trace, for the timeline (tools/kernel_timeline.py)
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_overlap -o bench -- python $R/bench.py --no-cpu-baseline --no-steady-state --steps 20 > /dev/null 2>&1
cd $R
python tools/kernel_timeline.py gpurun_out/prof_${tag}_overlap/bench_results.db 24 170 > gpurun_out/timeline_$tag.txt 2>&1
python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
# the driver's own command at round end (VERDICT r05 task 6: the round is not closed before it has been run on the final tree), with and without the second stream,
# with the working sets carried, and the batch sweep
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${tag}_driver.json 2> /dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-overlap --no-cpu-baseline > gpurun_out/bench_${tag}_driver_no_overlap.json 2> /dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --wbc-state carry > gpurun_out/bench_${tag}_driver_carry.json 2> /dev/null
python bench.py --no-cpu-baseline --no-overlap > gpurun_out/bench_${tag}_no_overlap.json 2> /dev/null
python bench.py --no-cpu-baseline --sweep > gpurun_out/bench_${tag}_sweep.json 2> /dev/null
python tools/adapter_latency.py > gpurun_out/adapter_latency_$tag.txt 2>&1
python tools/pcie_rate.py > gpurun_out/pcie_rate_$tag.txt 2>&1
python tools/wbc_tail_probe.py --steps 40 --above 18 > gpurun_out/wbc_tail_$tag.txt 2>&1
tail -1 gpurun_out/bench_$tag.json

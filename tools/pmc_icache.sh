#!/bin/bash
# usage: bash tools/pmc_icache.sh <label> [bench args...]  -- instruction-cache and scalar-cache counters of the bench kernels
L=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout -s KILL 180 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmci_${L}_1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmci_${L}_1.log 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/pmci_${L}_2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmci_${L}_2.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmci_${L}_1 $R/gpurun_out/pmci_${L}_2

#!/bin/bash
# usage: bash tools/pmc.sh <label> [bench args...]   -- collects two SQ counter passes of bench.py into gpurun_out/pmc_<label>_{1,2}
L=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_${L}_1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_${L}_1.log 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_${L}_2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_${L}_2.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${L}_1 $R/gpurun_out/pmc_${L}_2

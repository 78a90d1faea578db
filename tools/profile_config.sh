#!/bin/bash
# usage: bash tools/profile_config.sh <label> [bench.py args...]
# One kernel-trace pass and five counter passes (each its own run of bench.py) of one benchmark configuration:
#   gpurun_out/prof_<label>_trace   rocprofv3 --kernel-trace --stats
#   gpurun_out/prof_<label>_{1..5}  SQ (two passes), FETCH_SIZE + TCC_REQ, WRITE_SIZE, TCC hit/miss, TCP/TA
# and a summary on stdout (tools/pmc_summary.py).
L=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $R/gpurun_out/prof_${L}_trace $R/gpurun_out/prof_${L}_[1-5]   # results of earlier runs would be averaged in
cd /tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${L}_trace -- $BENCH > $R/gpurun_out/prof_${L}_trace.log 2>&1 || exit 1
echo "trace pass done"
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
P2="SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"
P3="FETCH_SIZE TCC_REQ_sum"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
P5="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  timeout -s KILL 300 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/prof_${L}_$i -- $BENCH > $R/gpurun_out/prof_${L}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/prof_${L}_$i.log; exit 1; }
  echo "pass $i done"
  i=$((i+1))
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/prof_${L}_1 $R/gpurun_out/prof_${L}_2 $R/gpurun_out/prof_${L}_3 $R/gpurun_out/prof_${L}_4 $R/gpurun_out/prof_${L}_5
find $R/gpurun_out/prof_${L}_trace -name "*kernel_stats.csv" | head -1 | xargs cat

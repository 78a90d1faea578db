#!/bin/bash
# usage: bash tools/pmc_bench.sh <label> <steps> [bench.py args...]
# Counter passes of one benchmark configuration at its stated size (each pass is a run of bench.py of its own), and a
# kernel trace:  gpurun_out/pmcb_<label>_trace  rocprofv3 --kernel-trace --stats
#                gpurun_out/pmcb_<label>_{1,2,3}  FETCH_SIZE + TCC_REQ, WRITE_SIZE + TCC hit / miss, SQ lane utilisation
# tools/pmc_bench_digest.py turns them into profiles/<label>_pmc.txt and an entry of bench_traffic.json.
L=$1; STEPS=$2; shift; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $R/gpurun_out/pmcb_${L}_trace $R/gpurun_out/pmcb_${L}_[1-3]
cd /tmp
BENCH="python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline $*"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcb_${L}_trace -- $BENCH > $R/gpurun_out/pmcb_${L}_trace.log 2>&1 || { echo "trace pass failed"; tail -3 $R/gpurun_out/pmcb_${L}_trace.log; exit 1; }
echo "$L: trace pass done"
P1="FETCH_SIZE TCC_REQ_sum"
P2="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
P3="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -s KILL 400 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmcb_${L}_$i -- $BENCH > $R/gpurun_out/pmcb_${L}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmcb_${L}_$i.log; exit 1; }
  echo "$L: pass $i done"
  i=$((i+1))
done

#!/bin/bash
# usage: bash tools/pmc_split.sh <label> <config> <spp> [library]
# HBM-side traffic and L2 behaviour of one render of a mesh configuration, for one library build:
# three counter passes (each its own run): FETCH_SIZE + TCC_REQ, WRITE_SIZE + TCC hit / miss, TCP requests.
L=$1; CFG=$2; SPP=$3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$4" ] && export SRT_LIBRARY=$R/$4
cd /tmp
P1="FETCH_SIZE TCC_REQ_sum"
P2="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
i=1
for P in "$P1" "$P2" "$P3"; do
  rm -rf $R/gpurun_out/pmcs_${L}_$i
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmcs_${L}_$i -- python3 $R/tools/big_scenes.py $CFG --spp $SPP --once > $R/gpurun_out/pmcs_${L}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmcs_${L}_$i.log; exit 1; }
  i=$((i+1))
done
echo "== $L"
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcs_${L}_1 $R/gpurun_out/pmcs_${L}_2 $R/gpurun_out/pmcs_${L}_3

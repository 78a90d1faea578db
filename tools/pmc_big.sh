#!/bin/bash
# usage: bash tools/pmc_big.sh <label> <config> <spp>
# Five counter passes (each its own run) over one render of a mesh configuration: gpurun_out/pmcbig_<label>_{1..5}
L=$1; CFG=$2; SPP=$3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
P2="SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"
P3="FETCH_SIZE TCC_REQ_sum"
P4="TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_EA0_RDREQ_sum"
P5="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmcbig_${L}_$i -- python3 $R/tools/big_scenes.py $CFG --spp $SPP --once > $R/gpurun_out/pmcbig_${L}_$i.log 2>&1 || exit 1
  i=$((i+1))
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcbig_${L}_1 $R/gpurun_out/pmcbig_${L}_2 $R/gpurun_out/pmcbig_${L}_3 $R/gpurun_out/pmcbig_${L}_4 $R/gpurun_out/pmcbig_${L}_5

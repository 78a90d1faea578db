#!/bin/bash
# usage (on the GPU box): bash tools/wf_pmc.sh <config> <label> [ENV=VALUE ...]   -- SQ counters per kernel of one ab.py child run
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$1; L=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/wfpmc_$L -- python3 $R/tools/ab.py --child $C 1 > $R/gpurun_out/wfpmc_$L.log 2>&1 || { echo "failed"; tail -3 $R/gpurun_out/wfpmc_$L.log; exit 1; }
python3 - "$R/gpurun_out/wfpmc_$L" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    if c.get("SQ_WAVE_CYCLES", 0) == 0: continue
    print("%-42s calls %4d busy %.3e valu_insts %.3e wave_cyc %.3e  valu_active/wave_cyc %.3f  wait_inst/wave_cyc %.3f  lanes/valu %.1f vmem %.3e" % (
        k, calls[k], c["SQ_BUSY_CYCLES"], c["SQ_INSTS_VALU"], c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
        c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1) , c["SQ_INSTS_VMEM"]))
PY

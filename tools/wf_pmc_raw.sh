#!/bin/bash
# usage (on the GPU box): bash tools/wf_pmc_raw.sh <config> <label> "<counter> <counter> ..."   -- per-kernel sums of any counters, one ab.py child run
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$1; L=$2; P=$3
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/wfraw_$L -- python3 $R/tools/ab.py --child $C 1 > $R/gpurun_out/wfraw_$L.log 2>&1 || { echo "failed"; tail -3 $R/gpurun_out/wfraw_$L.log; exit 1; }
python3 - "$R/gpurun_out/wfraw_$L" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-40:] + ("<shadow>" if "wf_walk<true" in r["Kernel_Name"] else "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
for k, c in sorted(agg.items()):
    print("%-44s calls %4d  " % (k, len(calls[k])) + "  ".join("%s %.4e" % (n, v) for n, v in sorted(c.items())))
PY

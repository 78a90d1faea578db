#!/bin/bash
# usage (on the GPU box): bash tools/wf_profile.sh <config> [label]   -- per-kernel times of one ab.py child run
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$1; L=${2:-$1}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wfprof_$L -- python3 $R/tools/ab.py --child $C 2 > $R/gpurun_out/wfprof_$L.log 2>&1 || { echo "failed"; tail -3 $R/gpurun_out/wfprof_$L.log; exit 1; }
python3 - "$R/gpurun_out/wfprof_$L" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print("  %-70s calls %6s total_ms %10.3f avg_us %10.1f pct %s" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY

#!/bin/bash
# usage: bash tools/pmc_traffic.sh <label> [bench args...] -- HBM-side bytes of the bench's kernels: FETCH_SIZE and WRITE_SIZE, one pass each
L=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmct_${L}_1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmct_${L}_1.log 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmct_${L}_2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmct_${L}_2.log 2>&1 || exit 1
python3 - $R/gpurun_out/pmct_${L}_1 $R/gpurun_out/pmct_${L}_2 <<'PY'
import collections, csv, glob, sys
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            tot[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print(k, {c: (len(x), sum(x) / len(x)) for c, x in v.items()})
PY

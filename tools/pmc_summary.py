"""Summarises rocprofv3 --pmc CSV output per kernel: python tools/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
for k, v in tot.items():
    if "trace_kernel" not in k:
        continue
    per = {c: v[c] / calls[k][c] for c in v}
    print(k)
    for c in sorted(per):
        print("   %-24s %.4g" % (c, per[c]))
    if "SQ_THREAD_CYCLES_VALU" in per and "SQ_ACTIVE_INST_VALU" in per:
        print("   lane utilisation (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU * 64)) = %.3f" % (per["SQ_THREAD_CYCLES_VALU"] / (per["SQ_ACTIVE_INST_VALU"] * 64)))
    if "SQ_WAVE_CYCLES" in per and "SQ_BUSY_CYCLES" in per:
        print("   WAVE_CYCLES / BUSY_CYCLES = %.3f" % (per["SQ_WAVE_CYCLES"] / per["SQ_BUSY_CYCLES"]))

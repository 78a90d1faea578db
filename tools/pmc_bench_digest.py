"""Digest of tools/pmc_bench.sh output (under gpurun_out/) for one configuration:

    python tools/pmc_bench_digest.py <label> <config> <steps>

writes profiles/<label>_pmc.txt (per-kernel counter averages per launch, the kernel trace's statistics) and adds the
configuration's entry to bench_traffic.json: counter sums PER FRAME (a frame of the big configurations is
several launches of the trace kernel: the counters are summed over the launches of the run and divided by its frames).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    label, config, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    base = os.path.join(ROOT, "gpurun_out", "pmcb_%s_" % label)
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for i in (1, 2, 3):
        for f in glob.glob(base + "%d/**/*counter_collection.csv" % i, recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k][r["Counter_Name"]] += 1
    lines = []
    main_kernel, main_calls = None, 0
    for k, v in tot.items():
        lines.append(k)
        for c in sorted(v):
            lines.append("   %-26s %.5g per launch (%d launches)" % (c, v[c] / calls[k][c], calls[k][c]))
        if "SQ_THREAD_CYCLES_VALU" in v and "SQ_ACTIVE_INST_VALU" in v:
            lines.append("   lane utilisation (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU * 64)) = %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
            lines.append("   VALU issue share of a wave's cycles = %.3f, waiting share = %.3f" % (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
        # the frame's trace kernel: the non-counting instantiation with the most launches
        args = k.split("<")[1].rstrip(">").split(",") if "trace_kernel<" in k else []
        if len(args) >= 2 and args[1].strip() == "false" and calls[k].get("FETCH_SIZE", 0) > main_calls:
            main_kernel, main_calls = k, calls[k]["FETCH_SIZE"]
    stats = ""
    for f in glob.glob(base + "trace/**/*kernel_stats.csv", recursive=True):
        stats = open(f).read()
    per_frame_frames = steps + 1      # frames rendered by the non-counting kernel in one run: warm-up + timed steps
    v = tot[main_kernel]
    entry = {"kernel": main_kernel, "launches_per_frame": int(round(calls[main_kernel]["FETCH_SIZE"] / per_frame_frames)),
             "fetch_kib": v["FETCH_SIZE"] / per_frame_frames, "write_kib": v["WRITE_SIZE"] / per_frame_frames,
             "tcc_req": v["TCC_REQ_sum"] / per_frame_frames, "tcc_hit": v["TCC_HIT_sum"] / per_frame_frames, "tcc_miss": v["TCC_MISS_sum"] / per_frame_frames,
             "lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64), 3),
             "source": "profiles/%s_pmc.txt" % label}
    with open(os.path.join(ROOT, "profiles", "%s_pmc.txt" % label), "w") as f:
        f.write("python bench.py --config %s --steps %d --warmup 1 --no-cpu-baseline under rocprofv3 (three --pmc passes, one --kernel-trace --stats pass)\n" % (config, steps))
        f.write("\n".join(lines) + "\n\n" + stats)
    path = os.path.join(ROOT, "bench_traffic.json")
    digest = json.load(open(path)) if os.path.exists(path) else {}
    digest[config] = entry
    json.dump(digest, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()

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
        # gpurun merges every run's files into the same directories: only the newest run's file counts
        for f in sorted(glob.glob(base + "%d/**/*counter_collection.csv" % i, recursive=True), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k][r["Counter_Name"]] += 1
    lines = []
    main_kernel, main_calls = None, 0
    for k, v in tot.items():
        lines.append(k)
        for c in sorted(v):
            lines.append("   %-26s %.5g per launch (%d launches)" % (c, v[c] / calls[k][c], calls[k][c]))
        if v.get("SQ_ACTIVE_INST_VALU", 0) > 0 and v.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_THREAD_CYCLES_VALU" in v:
            lines.append("   lane utilisation (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU * 64)) = %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
            lines.append("   VALU issue share of a wave's cycles = %.3f, waiting share = %.3f" % (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
        # the frame's trace kernel: the non-counting instantiation with the most launches
        args = k.split("<")[1].rstrip(">").split(",") if "trace_kernel<" in k else []
        if len(args) >= 2 and args[1].strip() == "false" and calls[k].get("FETCH_SIZE", 0) > main_calls:
            main_kernel, main_calls = k, calls[k]["FETCH_SIZE"]
    # Scenes with deep ray trees: the first sample of the warm-up frame goes through mesh_trace_kernel (the probe), everything
    # after it through the kernels of the level-by-level tracer (wf_*).  Their counters are summed over all of those kernels;
    # the counting render at the end (a 16-spp frame) shares the kernels that do no traversal, so it counts as 16 / spp of a
    # frame, and the warm-up frame as (spp - 1) / spp.
    wf = [k for k in tot if "wf_" in k and calls[k].get("FETCH_SIZE", 0) > 0]
    if any("wf_walk" in k for k in wf) and sum(calls[k]["FETCH_SIZE"] for k in wf) > 8 * main_calls:
        sys.path.insert(0, ROOT)
        spp = {"c3": 64, "c4": 256, "c5": 1024}.get(config, 16)
        frames = steps + (spp - 1.0) / spp + 16.0 / spp
        summed = collections.defaultdict(float)
        for k in wf:
            for c, x in tot[k].items():
                summed[c] += x
        walk = max((k for k in wf if "wf_walk<false" in k or "wf_walk<0" in k), key=lambda k: calls[k]["FETCH_SIZE"], default=wf[0])
        w = tot[walk]
        entry = {"kernel": "level-by-level tracer: " + ", ".join(sorted(set(k.split("::")[-1].split("<")[0] for k in wf))),
                 "launches_per_frame": int(round(sum(calls[k]["FETCH_SIZE"] for k in wf) / frames)),
                 "fetch_kib": summed["FETCH_SIZE"] / frames, "write_kib": summed["WRITE_SIZE"] / frames,
                 "tcc_req": summed["TCC_REQ_sum"] / frames, "tcc_hit": summed["TCC_HIT_sum"] / frames, "tcc_miss": summed["TCC_MISS_sum"] / frames,
                 "lane_utilisation": round(summed["SQ_THREAD_CYCLES_VALU"] / (summed["SQ_ACTIVE_INST_VALU"] * 64), 3),
                 "lane_utilisation_walkers": round(w["SQ_THREAD_CYCLES_VALU"] / (w["SQ_ACTIVE_INST_VALU"] * 64), 3),
                 "source": "profiles/%s_pmc.txt" % label}
        stats = ""
        for f in sorted(glob.glob(base + "trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1:]:
            stats = open(f).read()
        with open(os.path.join(ROOT, "profiles", "%s_pmc.txt" % label), "w") as f:
            f.write("python bench.py --config %s --steps %d --warmup 1 --no-cpu-baseline under rocprofv3 (three --pmc passes, one --kernel-trace --stats pass)\n" % (config, steps))
            f.write("\n".join(lines) + "\n\n" + stats)
        path = os.path.join(ROOT, "bench_traffic.json")
        digest = json.load(open(path)) if os.path.exists(path) else {}
        digest[config] = entry
        json.dump(digest, open(path, "w"), indent=1, sort_keys=True)
        print(json.dumps(entry))
        return
    stats = ""
    for f in sorted(glob.glob(base + "trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1:]:
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

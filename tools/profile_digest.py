"""Digest of tools/profile_config.sh output: python tools/profile_digest.py <label> <kernel name fragment> (files under profiles/)."""
import csv
import re
import sys

L2_PEAK_TBS, HBM_PEAK_TBS = 34.5, 8.0


def digest(label, name):
    txt = open("profiles/%s_pmc_and_kernel_stats.txt" % label).read()
    i = txt.index(name + "\n")
    j = txt.find("_kernel<", i + len(name))
    k = txt.find('"Name"', i)
    blk = txt[i:min(x for x in (j, k, len(txt)) if x > 0)]
    v = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\w+)\s+([\d.e+]+)$", blk, re.M)}
    ms = None
    for row in csv.DictReader(open("profiles/%s_kernel_stats.csv" % label)):
        if name in row["Name"]:
            ms = float(row["AverageNs"]) / 1e6
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024      # KiB; FETCH_SIZE counts half of the bytes on gfx950
    l2 = v["TCC_REQ_sum"] * 128 / (ms * 1e-3) / 1e12
    return {"kernel": name, "avg_ms": round(ms, 3),
            "valu_lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64), 3),
            "valu_issue_share_of_wave_cycles": round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 3),
            "wait_inst_share": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3), "wait_any_share": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3),
            "salu_per_valu": round(v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"], 3),
            "hbm_side_GB_per_launch": round(hbm / 1e9, 3), "hbm_side_TBs": round(hbm / (ms * 1e-3) / 1e12, 3), "hbm_frac": round(hbm / (ms * 1e-3) / 1e12 / HBM_PEAK_TBS, 4),
            "l2_request_TBs": round(l2, 2), "l2_frac": round(l2 / L2_PEAK_TBS, 4),
            "l2_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3),
            "l1_requests_per_l2_request": round(v["TCP_TOTAL_CACHE_ACCESSES_sum"] / max(1.0, v["TCP_TCC_READ_REQ_sum"]), 2)}


if __name__ == "__main__":
    import json
    print(json.dumps(digest(sys.argv[1], sys.argv[2])))

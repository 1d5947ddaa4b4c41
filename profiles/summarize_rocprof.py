#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into a small JSON/markdown.

usage: summarize_rocprof.py <kernel_stats.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_in_pmc_runs>
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so the read
side is doubled (upper estimate for narrow accesses).
"""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("srh::", "")
    return n.split("<")[0]


def pmc(path, counter):
    tot = defaultdict(float)
    cnt = defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def main():
    stats_csv, fetch_csv, write_csv = sys.argv[1:4]
    out = {}
    with open(stats_csv) as f:
        for row in csv.DictReader(f):
            out[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                                       "total_ms": float(row["TotalDurationNs"]) / 1e6, "pct": float(row["Percentage"])}
    ft, fc = pmc(fetch_csv, "FETCH_SIZE")
    wt, wc = pmc(write_csv, "WRITE_SIZE")
    for k in out:
        if k in ft:
            out[k]["pmc_launches"] = fc[k]
            out[k]["fetch_KiB_raw_per_launch"] = ft[k] / fc[k]
            out[k]["write_KiB_per_launch"] = wt.get(k, 0.0) / max(1, wc.get(k, 0))
            out[k]["hbm_bytes_per_launch_corrected"] = (2.0 * ft[k] / fc[k] + wt.get(k, 0.0) / max(1, wc.get(k, 0))) * 1024
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

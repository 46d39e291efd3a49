"""Summarise rocprofv3 output into profiles/rNN/ (per-kernel counter means, kernel-time stats).

usage: python tools/pmc_summary.py <dir with pass sub-directories> <out.json> [kernel_stats.csv]

Every `*_counter_collection.csv` and every rocpd `*_results.db` (rocprofv3's default output in
ROCm 7) below the directory is read; a dispatch's counter value is the sum over its rows (one
row per counter dimension), a kernel's figure the mean over its dispatches.  With a third
argument the kernel-trace database(s) are reduced to a `--stats`-style CSV (calls, total /
average / min / max ns per kernel).  HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B requests in KiB as if they were 32 B).
"""
import csv
import glob
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").strip()


def main():
    root, out = sys.argv[1], sys.argv[2]
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for path in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if not k.startswith("exabm4d"):
                    continue
                per[k][row["Counter_Name"]][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
    durations = defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True):
        db = sqlite3.connect(path)
        for name, disp, counter, value in db.execute(
                "select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            k = short(name)
            if k.startswith("exabm4d"):
                per[k][counter][(path, disp)] += float(value)
        if not per:
            for name, dur in db.execute("select name, duration from kernels"):
                durations[short(name)].append(int(dur))
        db.close()
    summary = {}
    for k, counters in sorted(per.items()):
        summary[k] = {c: {"launches": len(v), "per_launch_mean": sum(v.values()) / len(v)}
                      for c, v in sorted(counters.items())}
    with open(out, "w") as f:
        json.dump(summary, f, indent=1)
    if len(sys.argv) > 3 and durations:
        total = sum(sum(v) for v in durations.values())
        with open(sys.argv[3], "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for k, v in sorted(durations.items(), key=lambda kv: -sum(kv[1])):
                wr.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1),
                             round(100.0 * sum(v) / total, 3), min(v), max(v)])
                print(k, len(v), round(sum(v) / len(v) / 1e6, 3), "ms avg")
    for k, c in summary.items():
        print(k, {n: round(v["per_launch_mean"], 1) for n, v in c.items()})


if __name__ == "__main__":
    main()

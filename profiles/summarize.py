"""Condense rocprofv3 output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    print("%-70s %8s %14s %14s %8s" % ("Name", "Calls", "TotalNs", "AverageNs", "Pct"))
    for r in rows[:12]:
        print("%-70s %8s %14s %14.0f %8s" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"],
                                             float(r["AverageNs"]), r.get("Percentage", "")))

for f in find("*counter_collection.csv"):
    print("== counters:", os.path.relpath(f, out))
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
    for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print("%-60s %-12s dispatches=%d total=%.6g per_dispatch=%.6g" % (k, c, n, v, v / n))

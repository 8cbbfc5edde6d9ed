"""Summarise a gpurun_out/prof_<tag>/ directory: kernel stats + PMC counters per kernel (mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"),
                                                               row.get("AverageNs"), row.get("Percentage")))
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== pmc:", os.path.relpath(f, out))
        for k, cs in acc.items():
            if "trace_kernel" not in k and "reduce" not in k and "assemble" not in k:
                continue
            print("  ", k)
            for c, v in sorted(cs.items()):
                print("      %-28s mean/dispatch = %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))

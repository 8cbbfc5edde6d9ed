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

# traffic of the trace kernel for profiles/pmc_traffic.json: python scripts/summarize_prof.py <dir> --traffic KEY
if len(sys.argv) > 3 and sys.argv[2] == "--traffic":
    import json
    vals = {}
    for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "trace_kernel" in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    try:
        allv = json.load(open(path))
    except (OSError, ValueError):
        allv = {}
    allv[sys.argv[3]] = {"fetch_size_kb": sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]), "write_size_kb": sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]),
                         "source": os.path.basename(out.rstrip("/")), "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean per trace_kernel dispatch, KB; "
                         "uncorrected: gfx950 halves FETCH_SIZE only for wide coalesced streams, this kernel's reads are 64-byte node / scalar loads"}
    json.dump(allv, open(path, "w"), indent=1, sort_keys=True)
    print("updated", path, sys.argv[3], allv[sys.argv[3]])

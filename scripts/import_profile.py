#!/usr/bin/env python3
"""Copy a gpurun_out/prof_<tag>/ profile (scripts/gpu_prof.sh) into profiles/ under a round-stamped name and record the
trace kernel's per-launch PMC figures in profiles/pmc_counters.json (what bench.py's roofline object imports).

usage: python scripts/import_profile.py <prof dir> <name, e.g. round2_C3_bvh> <key, e.g. C3/bvh/f64>"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, name, key = sys.argv[1], sys.argv[2], sys.argv[3]
prof = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(prof, name + "_summary.txt"))
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(prof, name + "_kernel_stats.csv"))
bj = os.path.join(src, "bench_trace.json")
if os.path.exists(bj) and os.path.getsize(bj):
    shutil.copy(bj, os.path.join(prof, name + "_bench.json"))
vals = defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in vals.items()}
avg_ns = None
for row in csv.DictReader(open(stats)):
    if "trace_kernel" in row["Name"] and (avg_ns is None or int(row["Calls"]) > calls):
        avg_ns, calls = float(row["AverageNs"]), int(row["Calls"])
sys.path.insert(0, ROOT)
import bench
sha = bench.kernel_sha()
try:  # recorded on the GPU box when the profile was taken (scripts/gpu_prof.sh)
    sha = open(os.path.join(src, "kernel_sha.txt")).read().strip() or sha
except OSError:
    pass
launches_per_frame = 1
try:
    launches_per_frame = int(json.loads(open(bj).read().strip().splitlines()[-1])["roofline"]["launches_per_step"])
except (OSError, ValueError, KeyError, IndexError):
    pass
fp64 = sum(m.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
classes = {c: m.get("SQ_INSTS_VALU_" + c) for c in ("FMA_F32", "ADD_F32", "MUL_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "TRANS_F32", "INT32", "INT64", "CVT")}
entry = {
    "valu_classes": {c: v for c, v in classes.items() if v is not None},  # priced per class by bench.py (profiles/valu_prices.json)
    "valu_active_quad_cycles": m.get("SQ_ACTIVE_INST_VALU"), "wave_quad_cycles": m.get("SQ_WAVE_CYCLES"), "branch_total": m.get("SQ_INSTS_BRANCH"),
    "valu_total": m["SQ_INSTS_VALU"], "valu_fp64": fp64, "salu_total": m.get("SQ_INSTS_SALU"), "vmem_total": m.get("SQ_INSTS_VMEM"),
    "lanes_active": round(m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"]), 4),
    "wave_time_share": {"issuing": round(m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4), "waiting_for_issue": round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4),
                        "in_s_waitcnt": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4)},
    "fetch_kb": m["FETCH_SIZE"], "write_kb": m["WRITE_SIZE"], "rocprof_avg_launch_ms": round(avg_ns / 1e6, 4),
    "launches_per_frame": launches_per_frame,
    "source": "profiles/%s_summary.txt" % name, "kernel_sha": sha,
    "note": "rocprofv3 --pmc, one pass per counter group, mean per trace_kernel dispatch; FETCH_SIZE / WRITE_SIZE in KB, uncorrected (the kernel's "
            "reads are 32-byte node / scalar loads, not wide coalesced streams; the writes are 24-byte per-sample records)",
}
path = os.path.join(prof, "pmc_counters.json")
try:
    allv = json.load(open(path))
except (OSError, ValueError):
    allv = {}
allv[key] = entry
json.dump(allv, open(path, "w"), indent=1, sort_keys=True)
print(key, json.dumps(entry)[:400])

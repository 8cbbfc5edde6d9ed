#!/usr/bin/env python3
"""Where does make-final's time go?  Trace-kernel ms, ray segments, node visits and exact tests per segment of the scene (scene.clj:415-489) with parts of
it taken out (run on the GPU box): python scripts/gpu_final_variants.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytrace_clj_amd as r
from raytrace_clj_amd import core, flatten as fl

nx = ny = 500
ns = 128
ALL = ["ground", "light", "moving", "glass", "metal", "bndry", "medium", "haze", "earth", "marble", "cube"]
variants = [("all", ALL), ("no haze", [p for p in ALL if p != "haze"]), ("no media", [p for p in ALL if p not in ("haze", "medium")]),
            ("no cube", [p for p in ALL if p != "cube"]), ("no ground", [p for p in ALL if p != "ground"]),
            ("ground + light", ["ground", "light"]), ("cube + light", ["cube", "light"]),
            ("no media, no cube", [p for p in ALL if p not in ("haze", "medium", "cube")])]
for name, parts in variants:
    f = fl.flatten(r.scene.make_final(nx, ny, parts=parts))
    ctx = core.Context(0, timing=True)
    ctx.set_option("accel", 1)
    ds = core.DeviceScene(f, ctx=ctx)
    best, cnt = 1e9, None
    for k in range(4):
        lin, q, cnt = ds.render(nx, ny, ns)
        best = min(best, ctx.last_trace_ms()[0])
    ctx.set_option("count_traversal", 1)
    ds.render(nx, ny, ns)
    a, b = ctx.last_traversal_counters()
    seg = int(cnt[0])
    print("%-20s prims %5d  trace %7.3f ms  seg/sample %.3f  ps/segment %6.1f  node visits/seg %5.2f  exact tests/seg %5.2f" % (
        name, f.n_prims, best, seg / (nx * ny * ns), best * 1e9 / seg, a / 2 / seg, b / seg), flush=True)
    ds.close(); ctx.close()

#!/usr/bin/env python3
"""Traversal cost by bounce: renders C3 (or C2) at depth 0, 1, 2, 3, 50 with the counting instantiation and with the default kernel and
prints segments, node visits and exact tests per segment, and the launch time: differences between successive depths are the cost of
that bounce's segments (primary rays are coherent; bounced rays start ON a surface inside the tree)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import raytrace_clj_amd as r
from raytrace_clj_amd import core, flatten as fl

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
nx, ny, ns, n = {"C2": (800, 400, 64, 11), "C3": (1920, 1080, 64, 50)}[cfg]
scene = r.scene.make_random_scene(nx, ny, n, False, mix=(0.8, 0.95))
ctx = core.Context(0, timing=True)
ds = core.DeviceScene(fl.flatten(scene), ctx=ctx)
lin = torch.zeros((ny, nx, 3), dtype=torch.float64, device="cuda")
cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
prev = None
for depth in (0, 1, 2, 3, 5, 50):
    row = {}
    for count in (1, 0):
        ctx.set_option("count_traversal", count)
        for rep in range(2):
            ds.render_device(nx, ny, ns, lin, None, cnt, depth=depth)
            torch.cuda.synchronize()
            ms, launches = ctx.last_trace_ms()
        if count:
            a, b = ctx.last_traversal_counters()
            row.update(seg=int(cnt[0].item()), visits=a / 2, leaves=b)
        else:
            row.update(ms=ms)
    s = row["seg"]
    line = "depth %2d: segments %.4g (%.3f/sample)  visits/seg %.2f  exact/seg %.2f  launch %.3f ms" % (depth, s, s / (nx * ny * ns), row["visits"] / s, row["leaves"] / s, row["ms"])
    if prev:
        ds_, dv, dl, dm = s - prev["seg"], row["visits"] - prev["visits"], row["leaves"] - prev["leaves"], row["ms"] - prev["ms"]
        if ds_ > 0:
            line += "   | added segments %.4g: visits/seg %.2f exact/seg %.2f, %.2f ns/segment (vs %.2f for depth 0)" % (ds_, dv / ds_, dl / ds_, dm * 1e6 / ds_, first["ms"] * 1e6 / first["seg"])
    else:
        first = row
    print(line, flush=True)
    prev = row

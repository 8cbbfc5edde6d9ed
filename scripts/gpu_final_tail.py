#!/usr/bin/env python3
"""make-final's die-off: the spread of the workgroups' end times (diagnostic `make stamps` library: RTMI_LIB=.../librtmi_stamps.so) with and without the
subsurface sphere (a dielectric boundary around a dense medium: its paths run to depth 50).  python scripts/gpu_final_tail.py  (stderr carries the [stamps] lines)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytrace_clj_amd as r
from raytrace_clj_amd import core, flatten as fl

nx = ny = 500
ns = 128
ALL = ["ground", "light", "moving", "glass", "metal", "bndry", "medium", "haze", "earth", "marble", "cube"]
for name, parts in (("all", ALL), ("no subsurface sphere", [p for p in ALL if p not in ("medium", "bndry")]), ("no glass at all", [p for p in ALL if p not in ("medium", "bndry", "glass")])):
    f = fl.flatten(r.scene.make_final(nx, ny, parts=parts))
    ctx = core.Context(0, timing=True)
    ctx.set_option("accel", 1)
    ds = core.DeviceScene(f, ctx=ctx)
    print("== %s" % name, file=sys.stderr, flush=True)
    for k in range(2):
        ds.render(nx, ny, ns)
    print("%-24s trace %.3f ms" % (name, ctx.last_trace_ms()[0] / 2), flush=True)
    ds.close(); ctx.close()

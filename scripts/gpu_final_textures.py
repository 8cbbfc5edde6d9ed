#!/usr/bin/env python3
"""make-final with and without its two textured spheres (marble: Marble depth 4; earth: FlipTextureV of an ImageMap): trace ms and segments per sample --
what the textures cost beyond their rays' share of the segments is SIMD cost of one or two lanes evaluating a long texture (run on the GPU box; with
RTMI_LIB=.../librtmi_stamps.so the phase tables go to stderr)."""
import os, sys
sys.path.insert(0, os.getcwd())
import raytrace_clj_amd as r
from raytrace_clj_amd import core, flatten as fl, shader as shad, texture as tex
nx = ny = 500; ns = 128
ALL = ["ground", "light", "moving", "glass", "metal", "bndry", "medium", "haze", "earth", "marble", "cube"]
for name, parts in (("all", ALL), ("no marble", [p for p in ALL if p != "marble"]), ("no earth", [p for p in ALL if p != "earth"]), ("neither", [p for p in ALL if p not in ("earth", "marble")])):
    f = fl.flatten(r.scene.make_final(nx, ny, parts=parts))
    ctx = core.Context(0, timing=True); ctx.set_option("accel", 1)
    ds = core.DeviceScene(f, ctx=ctx)
    best = 1e9
    for k in range(6):
        lin, q, cnt = ds.render(nx, ny, ns); best = min(best, ctx.last_trace_ms()[0])
    print("%-12s trace %.3f ms  seg/sample %.3f" % (name, best, int(cnt[0]) / (nx * ny * ns)), flush=True)
    ds.close(); ctx.close()

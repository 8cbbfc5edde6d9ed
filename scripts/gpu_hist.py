#!/usr/bin/env python3
"""Lane-activity histograms of the BVH traversal (diagnostic build: make -C raytrace_clj_amd/csrc variant NAME=hist FLAGS=-DRTMI_HIST,
run with RTMI_LIB=.../librtmi_hist.so python scripts/gpu_hist.py C2|C3|FINAL).  The library prints, at context shutdown, for 0..64 active
lanes: HIST0 = inner-loop trips (node visits of a wave), HIST1 = exact-test phases, HIST2 = outer iterations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytrace_clj_amd as r
from raytrace_clj_amd import core, flatten as fl

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
if cfg == "FINAL":
    nx, ny, ns = 500, 500, 32
    scene = r.scene.make_final(nx, ny)
else:
    nx, ny, ns, n = {"C2": (800, 400, 64, 11), "C3": (1920, 1080, 32, 50)}[cfg]
    scene = r.scene.make_random_scene(nx, ny, n, False, mix=(0.8, 0.95))
ctx = core.Context(0)
ds = core.DeviceScene(fl.flatten(scene), ctx=ctx)
ds.render(nx, ny, ns)
ds.close()
ctx.close()

#!/usr/bin/env python3
"""Where does a tree start to pay?  Trace-kernel ms of small scenes under both accelerators (run on the GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytrace_clj_amd as r
from raytrace_clj_amd import core, flatten as fl

os.environ["RTMI_FLAT_BELOW"] = "0"  # measure what was asked for
scenes = [("cover%d" % n, r.scene.make_random_scene(800, 400, n, False), 800, 400, 64) for n in (1, 2, 3, 4, 5)]
scenes += [("cover%dm" % n, r.scene.make_random_scene(800, 400, n, True), 800, 400, 64) for n in (1, 2, 3)]
scenes += [("two-spheres", r.scene.make_two_spheres(800, 400), 800, 400, 64), ("two-perlin", r.scene.make_two_perlin_spheres(800, 400), 800, 400, 64),
           ("example-light", r.scene.make_example_light(800, 400), 800, 400, 64), ("subsurface", r.scene.make_subsurface_sphere(800, 400), 800, 400, 64), ("two-triangles", r.scene.make_two_triangles(800, 400), 800, 400, 64),
           ("cornell classic", r.scene.make_cornell_box(600, 600, True), 600, 600, 64), ("cornell smoke", r.scene.make_cornell_box(600, 600, False), 600, 600, 64)]
for name, sc, nx, ny, ns in scenes:
    f = fl.flatten(sc)
    ms = {}
    for accel in (1, 0):
        ctx = core.Context(0, timing=True)
        ctx.set_option("accel", accel)
        ds = core.DeviceScene(f, ctx=ctx)
        best = 1e9
        for k in range(4):
            out = ds.render(nx, ny, ns)
            best = min(best, ctx.last_trace_ms()[0])
        ms[accel] = best
        ds.close(); ctx.close()
    print("%-16s prims %4d   bvh %7.3f ms   flat %7.3f ms   flat/bvh %.2f" % (name, f.n_prims, ms[1], ms[0], ms[0] / ms[1]), flush=True)

#!/usr/bin/env python3
"""Soak test of the conservative float traversal (run on the GPU box: python scripts/gpu_soak.py [rays per scene]).

The BVH's float box tests are only a filter in front of the exact FP64 sphere test; their error budget (rtmi_device.h:
slab_hit / make_bvh_ray) must never lose a hit the exact Hitlist scan finds.  For many scenes (cover scenes at n = 3, 11, 50,
moving variants, random worlds with awkward radii / times / cameras) and millions of adversarial rays per scene (aimed at sphere
silhouettes within a few ulps, from near and far origins, plus random rays and rays along the path of real renders) the probe
results with accel = BVH must equal those with accel = flat bit for bit, in both precisions and both node formats."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytrace_clj_amd as r
from raytrace_clj_amd import core, flatten as fl
from test_gpu_parity import tangent_rays, random_rays, _random_scene, grazing_rays, layer_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scenes = [("cover3", r.scene.make_random_scene(200, 100, 3, False)), ("cover11", r.scene.make_random_scene(200, 100, 11, False)),
          ("cover11m", r.scene.make_random_scene(200, 100, 11, True)), ("cover50", r.scene.make_random_scene(200, 100, 50, False)),
          ("glass11", r.scene.make_random_scene(200, 100, 11, True, mix=(0.1, 0.2)))] + [("random%d" % s, _random_scene(s)) for s in range(8)]
bad = 0
# ---- the entry grid (bvh_grid_entry): per-cell trees of several shapes, layer scenes, grazing / border-aligned / surface rays ------------------
grid_scenes = [("cover11", scenes[1][1]), ("cover11m", scenes[2][1]), ("cover50", scenes[3][1]), ("layer0", layer_scene(0)), ("layer1", layer_scene(1, n=3000)),
               ("layer2", layer_scene(2, n=400, moving=False))]
for spec in ("1:4", "7:4", "48:4", "3:1", "16:2", "96:4"):
    os.environ["RTMI_GRID"], os.environ["RTMI_GRID_KMAX"] = spec.split(":")
    for name, sc in grid_scenes:
        f = fl.flatten(sc)
        ctx = core.Context(0)
        ds = core.DeviceScene(f, ctx=ctx)
        for prec in ("f64", "f32"):
            for chunk in range(0, n, 500_000):
                m = min(500_000, n - chunk)
                rays = np.concatenate([grazing_rays(f, m // 2, 4000 + chunk), tangent_rays(f, m // 4, 5000 + chunk), random_rays(m - m // 2 - m // 4, 6000 + chunk, spread=40.0)])
                for tmin in (0.001, 0.0):
                    ctx.set_option("accel", 0)
                    a = ds.probe_hit(rays, tmin, 3.4028234663852886e38, precision=prec)
                    ctx.set_option("accel", 1)
                    b = ds.probe_hit(rays, tmin, 3.4028234663852886e38, precision=prec)
                    diff = int((~np.all((a == b) | (np.isnan(a) & np.isnan(b)), axis=1)).sum())
                    bad += diff
                    if diff:
                        print("GRID MISMATCH", spec, name, prec, "tmin", tmin, diff, "of", len(rays))
        print("grid ok" if not bad else "grid BAD", spec, name, "prims", f.n_prims, "hit fraction %.3f" % float((b[:, 0] == 1).mean()), flush=True)
        ds.close(); ctx.close()
os.environ.pop("RTMI_GRID"); os.environ.pop("RTMI_GRID_KMAX")
for node16 in ("1", "0"):
    os.environ["RTMI_NODE16"] = node16
    for name, sc in scenes:
        f = fl.flatten(sc)
        ctx = core.Context(0)
        ds = core.DeviceScene(f, ctx=ctx)
        for prec in ("f64", "f32"):
            for chunk in range(0, n, 500_000):
                m = min(500_000, n - chunk)
                rays = np.concatenate([tangent_rays(f, m // 2, 1000 + chunk), random_rays(m // 4, 2000 + chunk), random_rays(m - m // 2 - m // 4, 3000 + chunk, spread=40.0)])
                for tmin in (0.001, 0.0):
                    ctx.set_option("accel", 0)
                    a = ds.probe_hit(rays, tmin, 3.4028234663852886e38, precision=prec)
                    ctx.set_option("accel", 1)
                    b = ds.probe_hit(rays, tmin, 3.4028234663852886e38, precision=prec)
                    diff = int((~np.all((a == b) | (np.isnan(a) & np.isnan(b)), axis=1)).sum())
                    bad += diff
                    if diff:
                        print("MISMATCH", name, prec, "node16", node16, "tmin", tmin, diff, "of", len(rays))
        print("ok" if not bad else "BAD", name, "node16=" + node16, "prims", f.n_prims, "hit fraction %.3f" % float((b[:, 0] == 1).mean()), flush=True)
        ds.close(); ctx.close()
# Renders: the time-sliced traversal (suspend_lanes 8, 3, 40) against the plain while-while instantiation (0) and the flat scan, whole images
# and counters, every scene, both node formats and precisions.
rbad = 0
# the entry grid in whole renders (time-sliced and plain instantiations) against the flat scan
for name, sc in grid_scenes:
    f = fl.flatten(sc)
    for prec in ("f64", "f32"):
        ctx = core.Context(0)
        ctx.set_option("accel", 0)
        ds = core.DeviceScene(f, ctx=ctx)
        ref = ds.render(320, 160, 12, precision=prec)
        ds.close(); ctx.close()
        for spec in ("0:4", "1:4", "7:2", "48:4"):
            os.environ["RTMI_GRID"], os.environ["RTMI_GRID_KMAX"] = spec.split(":")
            for lanes in (12, 0, 40):
                ctx = core.Context(0)
                ctx.set_option("suspend_lanes", lanes)
                ds = core.DeviceScene(f, ctx=ctx)
                out = ds.render(320, 160, 12, precision=prec)
                ds.close(); ctx.close()
                if not (np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]) and list(out[2]) == list(ref[2])):
                    rbad += 1
                    print("GRID RENDER MISMATCH", name, prec, spec, "suspend_lanes", lanes)
    print("grid renders ok" if not rbad else "grid renders BAD", name, flush=True)
os.environ.pop("RTMI_GRID"); os.environ.pop("RTMI_GRID_KMAX")
for node16 in ("1", "0"):
    os.environ["RTMI_NODE16"] = node16
    for name, sc in scenes:
        f = fl.flatten(sc)
        for prec in ("f64", "f32"):
            ref = None
            for accel, lanes in ((0, 0), (1, 0), (1, 8), (1, 3), (1, 40)):
                ctx = core.Context(0)
                ctx.set_option("accel", accel)
                ctx.set_option("suspend_lanes", lanes)
                ds = core.DeviceScene(f, ctx=ctx)
                out = ds.render(320, 160, 12, precision=prec)
                ds.close(); ctx.close()
                if ref is None:
                    ref = out
                elif not (np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]) and list(out[2]) == list(ref[2])):
                    rbad += 1
                    print("RENDER MISMATCH", name, prec, "node16", node16, "accel", accel, "suspend_lanes", lanes)
    print("renders ok" if not rbad else "renders BAD", "node16=" + node16, flush=True)
print("soak:", "PASS" if bad == 0 and rbad == 0 else "FAIL (%d rays, %d renders differ)" % (bad, rbad))
sys.exit(1 if bad or rbad else 0)

#!/usr/bin/env python3
"""The piecewise walk of long segments (RTMI_GRID_CHUNK) on scenes built to stress it: trace-kernel ms with the walk and with RTMI_GRID_WALK=0
(long segments from the root of the whole tree), images compared.  Run on the GPU box: python scripts/gpu_walk_cases.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytrace_clj_amd as r
from raytrace_clj_amd import core, flatten as fl
from raytrace_clj_amd.util import vec3


def layer(n, ext, cam_y, look_y, seed=1, rad=0.2):
    rng = np.random.default_rng(seed)
    H, S, T = r.hitable, r.shader, r.texture
    mat = S.lambertian(albedo=T.constant(color=vec3(0.5, 0.5, 0.5)))
    items = [H.uv_sphere(center=vec3(0, 0, 0), radius=5000, material=S.diffuse_light(tex=T.constant(color=vec3(0.7, 0.8, 1.0)))),
             H.sphere(center=vec3(0, -1000, 0), radius=1000, material=mat)]
    for _ in range(n):
        items.append(H.sphere(center=vec3(rng.uniform(-ext, ext), rad, rng.uniform(-ext, ext)), radius=rad, material=mat))
    cam = r.camera.thin_lens_camera(lookfrom=vec3(ext * 0.9, cam_y, ext * 0.3), lookat=vec3(0, look_y, 0), vup=vec3(0, 1, 0), vfov=40, aspect=2.0, aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0)
    return {"camera": cam, "world": H.hitlist(items=items)}


cases = [("sparse layer, camera inside it (400 spheres over 200 x 200, eye at y = 0.25 looking level)", layer(400, 100.0, 0.25, 0.25), 800, 400, 16),
         ("sparse layer, camera above (400 over 200 x 200, eye at y = 3)", layer(400, 100.0, 3.0, 0.0), 800, 400, 16),
         ("dense layer, camera inside it (10 000 over 100 x 100, eye at y = 0.25 looking level)", layer(10000, 50.0, 0.25, 0.25), 800, 400, 16),
         ("cover scene n = 50, its own camera", r.scene.make_random_scene(800, 400, 50, False), 800, 400, 16),
         ("cover scene n = 150 (90 000 spheres)", r.scene.make_random_scene(800, 400, 150, False), 800, 400, 16)]
for name, sc, nx, ny, ns in cases:
    f = fl.flatten(sc)
    res = {}
    for walk in ("1", "0"):
        os.environ["RTMI_GRID_WALK"] = walk
        ctx = core.Context(0, timing=True)
        ds = core.DeviceScene(f, ctx=ctx)
        best = 1e9
        for _ in range(3):
            out = ds.render(nx, ny, ns)
            best = min(best, ctx.last_trace_ms()[0])
        ctx.set_option("count_traversal", 1)
        o2 = ds.render(nx, ny, ns)
        trav = ctx.last_traversal_counters()
        res[walk] = (best, out, trav[0] / 2 / float(out[2][0]))
        ds.close(); ctx.close()
    same = all(np.array_equal(a, b) for a, b in zip(res["1"][1], res["0"][1]))
    print("%-95s prims %6d  walk %8.3f ms (%.2f visits/seg)  root %8.3f ms (%.2f)  identical %s" % (name, f.n_prims, res["1"][0], res["1"][2], res["0"][0], res["0"][2], same), flush=True)

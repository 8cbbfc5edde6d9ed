"""experiment: consecutive frames on two contexts / two streams so that frame k+1 fills the die-off tail of frame k"""
import sys
import time

import torch

import raytrace_clj_amd as r
from raytrace_clj_amd import dist as rdist

nx, ny, ns, steps = 800, 400, 64, 20
scene = r.scene.make_random_scene(nx, ny, 11, False)
flat = r.flatten.flatten(scene)
for depth in (1, 2, 3):
    sets = []
    for k in range(depth):
        ctx = r.Context(0, timing=False)
        ctx.set_option("accel", 1)
        ds = r.DeviceScene(flat, ctx=ctx)
        sets.append((ctx, ds, rdist.TileRenderer(ds, nx, ny, 0, 1), torch.cuda.Stream()))
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            ctx, ds, tr, st = sets[k % depth]
            with torch.cuda.stream(st):
                tr.step(ns)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print("pipeline depth %d: %.3f ms/frame, %.0f Msamples/s" % (depth, dt * 1e3, nx * ny * ns / dt / 1e6), flush=True)
    imgs = [s[2].rgb8.clone() for s in sets]
    assert all(torch.equal(imgs[0], i) for i in imgs)
    for ctx, ds, tr, st in sets:
        ds.close(); ctx.close()

#!/usr/bin/env python3
"""Soak test of the mixed-kind intersectors (run on the GPU box: python scripts/gpu_soak_ext.py [rays per scene] [scenes]).

Random worlds of boxes, rectangles, triangles, spheres and moving spheres under Translate / RotateY / FlipNormals wrappers -- as a bvh-node tree (the device's own
tree, leaf records, time-sliced traversal), as a Hitlist of at most 64 primitives (the small-world scan) and the same Hitlist with the small scan switched off
(the culled scan) -- probed with rays chosen to hurt: through rectangle corners and edges exactly, starting ON rectangle planes, axis-parallel (a zero direction
component: the refined reciprocal's plain-division path), towards sphere silhouettes, from inside boxes, with huge and tiny direction lengths, plus random rays.
Hitable.hit? of every ray must be the same bits on every device path, and (a sample of the rays) the same as the nested CPU oracle's for Hitlist worlds and
worlds without exact ties."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytrace_clj_amd as r
from raytrace_clj_amd import core

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
n_scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
H, S, T = r.hitable, r.shader, r.texture
vec3 = lambda *a: np.array(a, np.float64)


def world(seed, n_items):
    rng = np.random.default_rng(1000 + seed)
    mats = [S.lambertian(albedo=T.constant(color=vec3(*rng.random(3)))) for _ in range(5)]
    items, anchors = [], []
    for k in range(n_items):
        m = mats[k % 5]
        c = np.round(rng.normal(0, 12, 3) * 4) / 4  # quarter-unit coordinates: rays can meet edges and corners exactly
        kind = int(rng.integers(0, 6))
        if kind == 0:
            ext = np.round(1 + 4 * rng.random(3))
            o = H.box(p0=c, p1=c + ext, material=m)
            anchors += [c, c + ext, c + ext * vec3(1, 0, 0), c + 0.5 * ext]
        elif kind == 1:
            w, h = np.round(1 + 5 * rng.random(2))
            ax = int(rng.integers(0, 3))
            if ax == 0:
                o = H.rect_xy(x0=c[0], y0=c[1], x1=c[0] + w, y1=c[1] + h, k=c[2], material=m)
                anchors += [c, c + vec3(w, h, 0), c + vec3(w, 0, 0), c + vec3(0.5 * w, 0.5 * h, 0)]
            elif ax == 1:
                o = H.rect_xz(x0=c[0], z0=c[2], x1=c[0] + w, z1=c[2] + h, k=c[1], material=m)
                anchors += [c, c + vec3(w, 0, h), c + vec3(0, 0, h)]
            else:
                o = H.rect_yz(y0=c[1], z0=c[2], y1=c[1] + w, z1=c[2] + h, k=c[0], material=m)
                anchors += [c, c + vec3(0, w, h), c + vec3(0, w, 0)]
        elif kind == 2:
            o = H.triangle(v0=c, v1=c + np.round(rng.normal(0, 3, 3)), v2=c + np.round(rng.normal(0, 3, 3)), material=m)
            anchors += [c]
        elif kind == 3:
            rad = float(np.round(1 + 3 * rng.random(), 2))
            o = H.sphere(center=c, radius=rad, material=m)
            anchors += [c + vec3(rad, 0, 0), c + vec3(0, rad, 0), c]
        elif kind == 4:
            o = H.uv_sphere(center=c, radius=1.5, material=m)
            anchors += [c + vec3(0, 0, 1.5)]
        else:
            o = H.moving_sphere(center0=c, t0=0.0, center1=c + vec3(0, 1, 0), t1=1.0, radius=1.0, material=m)
            anchors += [c]
        w = int(rng.integers(0, 5))
        if w == 1:
            o = H.translate(item=o, offset=np.round(rng.normal(0, 5, 3)))
        elif w == 2:
            o = H.rotate_y(item=o, theta=float(rng.choice([15.0, -18.0, 90.0, 180.0, 37.5])))
        elif w == 3:
            o = H.translate(item=H.rotate_y(item=H.flip_normals(item=o), theta=float(rng.uniform(-90, 90))), offset=np.round(rng.normal(0, 5, 3)))
        elif w == 4:
            o = H.flip_normals(item=o)
        items.append(o)
    return items, np.array(anchors)


def rays_for(seed, anchors, n):
    rng = np.random.default_rng(2000 + seed)
    out = np.zeros((n, 7))
    k = n // 8
    a = anchors[rng.integers(0, len(anchors), n)]
    # 1. random origins aimed at anchors (corners, edges, centres, silhouettes): exact hits of the special points for axis-aligned offsets
    o = a + np.round(rng.normal(0, 20, (n, 3)))
    out[:, :3], out[:, 3:6] = o, a - o
    # 2. axis-parallel rays (one or two zero direction components) through anchors
    ax = rng.integers(0, 3, k)
    d = np.zeros((k, 3)); d[np.arange(k), ax] = rng.choice([-1.0, 1.0, 0.5, -3.0], k)
    out[:k, 3:6] = d
    out[:k, :3] = a[:k] - d * np.round(rng.uniform(1, 30, (k, 1)))
    # 3. origins ON the anchors (t = 0 candidates) with random directions
    out[k:2 * k, :3] = a[k:2 * k]
    out[k:2 * k, 3:6] = rng.normal(0, 1, (k, 3))
    # 4. huge / tiny direction lengths
    out[2 * k:3 * k, 3:6] *= 10.0 ** rng.integers(-12, 13, (k, 1))
    # 5. a zero direction, non-finite components (the float traversal's fallbacks)
    out[3 * k:3 * k + 8, 3:6] = 0.0
    out[3 * k + 8:3 * k + 16, 3] = np.inf
    out[3 * k + 16:3 * k + 24, 0] = np.nan
    out[:, 6] = rng.random(n)
    bad = ~np.isfinite(out[:, 3:6]).all(axis=1)
    out[bad & (np.arange(n) >= 3 * k + 24), 3:6] = 1.0
    return out


def probe(f, rays, accel, env=None, tmax=3.4028234663852886e38):
    for kv in (env or {}).items():
        os.environ[kv[0]] = kv[1]
    try:
        ctx = core.Context(0)
        ds = core.DeviceScene(f, ctx=ctx)
        ctx.set_option("accel", accel)
        out = ds.probe_hit(rays, t_max=tmax)
        ds.close(); ctx.close()
    finally:
        for kk in (env or {}):
            del os.environ[kk]
    return out


from oracle.oracle import Oracle  # noqa: E402  (the checker)
from oracle.tree import flatten_with_tree  # noqa: E402

oracle = Oracle("f64")
cam = r.camera.pinhole_camera(lookfrom=vec3(0, 0, 40), lookat=vec3(0, 0, 0), vup=vec3(0, 1, 0), vfov=40, aspect=1.0)
bad = 0
for s in range(n_scenes):
    small = s % 2 == 0
    items, anchors = world(s, 9 if small else 60)  # 9 items: at most 54 primitives (boxes are six) -- the small-world scan
    rays = rays_for(s, anchors, n_rays)
    f_list = flatten_with_tree({"camera": cam, "world": H.hitlist(items=items)})
    for form in ("hitlist", "bvh"):
        w = H.hitlist(items=items) if form == "hitlist" else H.make_bvh(items, 0.0, 1.0)
        f = flatten_with_tree({"camera": cam, "world": w})
        for tmax in (3.4028234663852886e38, 25.0):
            got = {"tree": probe(f, rays, 1, tmax=tmax), "scan": probe(f, rays, 0, tmax=tmax), "culled": probe(f, rays, 0, {"RTMI_SMALL_SCAN": "0"}, tmax=tmax),
                   "tree-noleafrec-box": probe(f, rays, 1, {"RTMI_BOX_LEAF": "1"}, tmax=tmax)}
            ref = got["tree"]
            for k, v in got.items():
                same = np.array_equal(v, ref, equal_nan=True)
                if not same:
                    bad += 1
                    d = np.where(~((v == ref) | (np.isnan(v) & np.isnan(ref))).all(axis=1))[0]
                    print("MISMATCH scene %d %s tmax %g: %s vs tree at %d rays, first %d: %s | %s" % (s, form, tmax, k, len(d), d[0], v[d[0], :4], ref[d[0], :4]), flush=True)
            # the oracle evaluates the nested records the way the reference does.  For a bvh-node world that includes what the device does NOT reproduce, on a set of
            # rays of measure zero that this script aims at deliberately: AABB.hit? is strict ((> tmax tmin), hitable.clj:47) and divides by zero direction
            # components (NaN for an origin in the plane), so a ray that touches a node's box in one point, runs inside one of its faces or is axis-parallel misses
            # the node and whatever it holds; and exact ties between siblings go to the right child (hitable.clj:103-105).  The device's contract for a bvh world is
            # the closest hit of the Hitlist scan over the same primitives: checked against the oracle's answer for the Hitlist of the same items (hit?, t, p,
            # normal, uv; not the index: the two forms number their primitives differently).
            m = min(len(rays), 20000)
            cols = list(range(9)) if form == "hitlist" else [0] + list(range(2, 9))
            exp = oracle.probe_hit(f if form == "hitlist" else f_list, rays[:m], tmax=tmax)
            ok = (ref[:m][:, cols] == exp[:, cols]) | (np.isnan(ref[:m][:, cols]) & np.isnan(exp[:, cols]))
            d = np.where(~ok.all(axis=1))[0]
            if len(d):
                bad += 1
                print("ORACLE MISMATCH scene %d %s tmax %g at %d rays, first %d: device %s oracle %s ray %s" % (s, form, tmax, len(d), d[0], ref[d[0], :4], exp[d[0], :4], rays[d[0]]), flush=True)
            print("scene %2d %-7s %2d items %3d prims tmax %-8g hits %.3f  ok" % (s, form, len(items), f.n_prims, tmax, ref[:, 0].mean()), flush=True)
print("SOAK-EXT %s (%d scenes x 2 forms x 2 intervals, %d rays each)" % ("PASS" if bad == 0 else "FAIL: %d" % bad, n_scenes, n_rays))
sys.exit(1 if bad else 0)

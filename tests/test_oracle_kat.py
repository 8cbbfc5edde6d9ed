"""Pins the CPU oracle (oracle/rt_oracle.c) with the known-answer data the reference's own tests hold
(test/raytrace_clj/util_test.clj, test/raytrace_clj/hitable_test.clj) and with analytic values derived
from the cited formulas."""
import itertools
import math

import numpy as np
import pytest

import raytrace_clj_amd as r
from raytrace_clj_amd import flatten as fl
from raytrace_clj_amd.util import vec3

FLT_MAX = 3.4028234663852886e38
MATERIAL = r.shader.lambertian(albedo=r.texture.constant(color=vec3(0.8, 0.8, 0.8)))  # hitable_test.clj:21

# hitable_test.clj:8-19
GRIDPOINTS = [25.0 * vec3(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)]
DIRECTIONS = [5.0 * vec3(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1) if (i, j, k) != (0, 0, 0)]


def world_of(*items):
    return fl.flatten(r.hitable.hitlist(items=list(items)), r.camera.PinholeCamera(*(np.zeros(3),) * 4))


def ray7(o, d, t=0.0):
    return np.concatenate([o, d, [t]])


# ---- K1: util_test.clj:44-49 ---------------------------------------------------------------------------
def test_k1_point_at_parameter(oracle):
    o, d = vec3(1, 2, 3), vec3(4, 5, 6)
    assert np.array_equal(oracle.point_at_parameter(o, d, 0), vec3(1, 2, 3))
    assert np.array_equal(oracle.point_at_parameter(o, d, 1), vec3(5, 7, 9))
    assert np.array_equal(oracle.point_at_parameter(o, d, -1), vec3(-3, -3, -3))


# ---- K2/K3: hitable_test.clj:23-59 ---------------------------------------------------------------------
def test_k2_sphere_lattice(oracle):
    for origin in GRIDPOINTS:
        s = r.hitable.sphere(center=origin, radius=1.0, material=MATERIAL)
        w = world_of(s)
        inward = np.array([ray7(origin + d, -d, 0.0) for d in DIRECTIONS])
        outward = np.array([ray7(origin + d, d, 0.1) for d in DIRECTIONS])
        assert oracle.probe_hit(w, inward, 0.0, FLT_MAX)[:, 0].all(), "intersect ray"
        assert not oracle.probe_hit(w, outward, 0.0, FLT_MAX)[:, 0].any(), "non-intersecting ray"
        graze = np.array([ray7(origin + vec3(1, 1, 0), vec3(-1, 0, 0)), ray7(origin + vec3(1, 1, 0), vec3(0, -1, 0)),
                          ray7(origin + vec3(1, 0, 1), vec3(0, 0, -1)), ray7(origin, vec3(1, 1, 1))])
        assert oracle.probe_hit(w, graze, 0.0, FLT_MAX)[:, 0].all(), "grazing rays x/y/z and ray from inside"


def test_k3_sphere_bbox(oracle):
    for origin in GRIDPOINTS:
        w = world_of(r.hitable.sphere(center=origin, radius=1.0, material=MATERIAL))
        vmin, vmax = oracle.prim_bbox(w, 0, 0, 0)
        assert np.array_equal(vmin, origin - 1.0) and np.array_equal(vmax, origin + 1.0)
        b = r.hitable.sphere(center=origin, radius=1.0, material=MATERIAL).bbox(0, 0)  # host mirror, same data
        assert np.array_equal(b.vmin, origin - 1.0) and np.array_equal(b.vmax, origin + 1.0)


# ---- K4/K5: hitable_test.clj:61-103 ---------------------------------------------------------------------
def test_k4_moving_sphere(oracle):
    t0, t1 = 0.1, 0.9
    for origin in GRIDPOINTS:
        s = r.hitable.moving_sphere(center0=origin, t0=t0, center1=origin + vec3(10, 20, 30), t1=t1, radius=1.0, material=MATERIAL)
        w = world_of(s)
        inward = np.array([ray7(origin + d, -d, t0) for d in DIRECTIONS])
        outward = np.array([ray7(origin + d, d, t0) for d in DIRECTIONS])
        assert oracle.probe_hit(w, inward, 0.0, FLT_MAX)[:, 0].all()
        assert not oracle.probe_hit(w, outward, 0.0, FLT_MAX)[:, 0].any()
        assert oracle.probe_hit(w, ray7(origin, vec3(1, 1, 1), t0), 0.0, FLT_MAX)[0, 0] == 1, "ray from inside"
        vmin, vmax = oracle.prim_bbox(w, 0, t0, t0)
        assert np.array_equal(vmin, origin - 1.0) and np.array_equal(vmax, origin + 1.0)
        b = s.bbox(t0, t0)
        assert np.array_equal(b.vmin, origin - 1.0) and np.array_equal(b.vmax, origin + 1.0)


def test_k5_center_at_time(oracle):
    pa, pb = vec3(0, 0, 0), vec3(1, 2, 3)
    assert np.array_equal(oracle.center_at_time(pa, 0, pb, 1, 0), pa)
    assert np.array_equal(oracle.center_at_time(pa, 0, pb, 1, 1), pb)
    assert np.array_equal(oracle.center_at_time(pa, 0, pb, 1, 0.5), vec3(0.5, 1.0, 1.5))
    assert np.array_equal(r.hitable.center_at_time(pa, 0, pb, 1, 0.5), vec3(0.5, 1.0, 1.5))


# ---- K6/K7: hitable_test.clj:106-141 -------------------------------------------------------------------
def test_k6_aabb(oracle):
    a, b = vec3(-1, -1, -1), vec3(1, 1, 1)
    cases = [(vec3(0, 0, 0), vec3(1, 1, 1)), (vec3(-2, 0, 0), vec3(1, 0, 0)), (vec3(0, -2, 0), vec3(0, 1, 0)),
             (vec3(0, 0, -2), vec3(0, 0, 1)), (vec3(-2, 1, 0), vec3(1, 0, 0)), (vec3(1, -2, 0), vec3(0, 1, 0)),
             (vec3(1, 0, -2), vec3(0, 0, 1))]
    for o, d in cases:
        assert oracle.aabb_hit(a, b, o, d, 0.0, FLT_MAX), (o, d)


def test_k7_surrounding_bbox(oracle):
    w = world_of(r.hitable.sphere(center=vec3(-1, 2, -3), radius=0.1, material=MATERIAL),
                 r.hitable.sphere(center=vec3(1, -2, 3), radius=0.1, material=MATERIAL))
    vmin, vmax = oracle.surrounding_bbox(oracle.prim_bbox(w, 0), oracle.prim_bbox(w, 1))
    assert list(vmin) == [-1.1, -2.1, -3.1] and list(vmax) == [1.1, 2.1, 3.1]
    hb = r.hitable.make_surrounding_bbox(r.hitable.sphere(center=vec3(-1, 2, -3), radius=0.1, material=None).bbox(0, 0),
                                         r.hitable.sphere(center=vec3(1, -2, 3), radius=0.1, material=None).bbox(0, 0))
    assert list(hb.vmin) == [-1.1, -2.1, -3.1] and list(hb.vmax) == [1.1, 2.1, 3.1]


# ---- K8: analytic sphere hits (hitable.clj:180-207) -----------------------------------------------------
def test_k8_sphere_analytic(oracle):
    w = world_of(r.hitable.sphere(center=vec3(0, 0, 0), radius=1.0, material=MATERIAL))
    h = oracle.probe_hit(w, ray7(vec3(0, 0, -5), vec3(0, 0, 1)), 0.0, FLT_MAX)[0]
    assert h[0] == 1 and h[2] == 4.0 and list(h[3:6]) == [0, 0, -1] and list(h[6:9]) == [0, 0, -1]
    h = oracle.probe_hit(w, ray7(vec3(0, 0, 0), vec3(1, 1, 1)), 0.0, FLT_MAX)[0]
    assert h[2] == pytest.approx(0.5773502691896257, abs=1e-15)
    assert np.allclose(h[6:9], 0.5773502691896257, atol=1e-15)
    h = oracle.probe_hit(w, ray7(vec3(1, 1, 0), vec3(-1, 0, 0)), 0.0, FLT_MAX)[0]  # discriminant exactly 0
    assert h[0] == 1 and h[2] == 1.0 and list(h[6:9]) == [0, 1, 0]
    # strict interval: t == t-min is rejected (hitable.clj:195)
    assert oracle.probe_hit(w, ray7(vec3(0, 0, -5), vec3(0, 0, 1)), 4.0, FLT_MAX)[0, 2] == 6.0
    assert oracle.probe_hit(w, ray7(vec3(0, 0, -5), vec3(0, 0, 1)), 0.0, 4.0)[0, 0] == 0


# ---- K9/K10: shader.clj:6-20, 69-74 -------------------------------------------------------------------
def test_k9_schlick(oracle):
    assert oracle.schlick(1.0, 1.5) == 0.04000000000000001
    assert oracle.schlick(0.0, 1.5) == 1.0
    assert oracle.schlick(0.5, 1.5) == pytest.approx(0.07, abs=1e-16)


def test_k10_reflect_refract(oracle):
    assert list(oracle.reflect(vec3(1, -1, 0), vec3(0, 1, 0))) == [1, 1, 0]
    assert np.allclose(oracle.refract(vec3(0, -1, 0), vec3(0, 1, 0), 1 / 1.5), [0, -1, 0], atol=1e-16)
    assert oracle.refract(vec3(1, -0.01, 0), vec3(0, 1, 0), 1.5) is None  # total internal reflection


# ---- K11: sky dome uv / emission (hitable.clj:128-139, texture.clj:26-34, scene.clj:336-344) --------------
def test_k11_dome(oracle, cover11):
    w = fl.flatten(cover11)
    dome = int(np.flatnonzero(w.prim_kind == fl.PRIM_UVSPHERE)[0])  # leaf order is the bvh's, not the list's
    assert w.prim_geom[dome, 3] == 1000
    dome_tex = int(w.mat_tex[w.prim_mat[dome]])
    cases = [((0, 1, 0), (0.5, 1.0), (1, 1, 1), (255, 255, 255)),
             ((1, 0, 0), (0.5, 0.5), (0.75, 0.85, 1), (221, 236, 255)),
             ((0, -1, 0), (0.5, 0.0), (0.5, 0.7, 1), (181, 214, 255)),
             ((0, 0, 1), (0.25, 0.5), (0.75, 0.85, 1), None),
             ((0.6, 0.8, 0), (0.5, 0.795167), (0.897584, 0.93855, 1), (242, 248, 255))]
    for n, uv, rgb, q in cases:
        got_uv = oracle.sphere_uv(vec3(*n))
        assert np.allclose(got_uv, uv, atol=1e-6)
        col = oracle.probe_texture(w, dome_tex, np.concatenate([got_uv, [0, 0, 0]]))[0]
        assert np.allclose(col, rgb, atol=1e-6)
        if q:
            assert tuple(oracle.quantise(col)) == q


# ---- K12: cover camera at aspect 2 (camera.clj:50-66, scene.clj:321-330) ------------------------------------
def test_k12_camera(oracle, cover11):
    c = fl.flatten(cover11).cam
    exp = dict(w=(0.963624111659, 0.148249863332, 0.222374794998), u=(0.224859506699, 0, -0.974391195695),
               v=(-0.144453361594, 0.988949937066, -0.033335391137), lleft=(2.82549317644, -1.226284198068, 4.271260490031),
               horiz=(1.585951915991, 0, -6.87245830263), vert=(-0.509420502061, 3.487571129492, -0.117558577399))
    assert np.allclose(c[18:21], exp["w"], atol=1e-11) and np.allclose(c[12:15], exp["u"], atol=1e-11)
    assert np.allclose(c[15:18], exp["v"], atol=1e-11) and np.allclose(c[3:6], exp["lleft"], atol=1e-11)
    assert np.allclose(c[6:9], exp["horiz"], atol=1e-11) and np.allclose(c[9:12], exp["vert"], atol=1e-11)
    # the oracle's own ctor (same formulas in C) agrees bit for bit with the host mirror
    oc = oracle.make_camera(1, vec3(13, 2, 3), vec3(0, 0, 0), vec3(0, 1, 0), 20, 2.0, 0.0, 10.0, 0.0, 1.0)
    assert np.array_equal(oc, c)
    # centre ray = -10 w; thin lens consumes disk draws (>= 2) and one time draw even at aperture 0
    ray = oracle.probe_camera(fl.flatten(cover11), [0.5, 0.5], [12345])[0]
    assert np.allclose(ray[3:6], (-9.636241116594, -1.482498633322, -2.223747949983), atol=1e-10)
    assert np.array_equal(ray[0:3], [13, 2, 3]) and ray[7] >= 3 and (ray[7] - 1) % 2 == 0 and 0 <= ray[6] < 1


# ---- K13: integrator (core.clj:17-41) -------------------------------------------------------------------
def _dome():
    return r.hitable.uv_sphere(center=vec3(0, 0, 0), radius=1000, material=r.shader.diffuse_light(
        tex=r.texture.uv_gradient(co=vec3(1, 1, 1), cu=vec3(1, 1, 1), cv=vec3(0.5, 0.7, 1.0), cuv=vec3(0.5, 0.7, 1.0))))


def test_k13_integrator(oracle):
    # dome-only world: every sample = emitted(dome), exactly one segment
    w = world_of(_dome())
    rays = np.array([ray7(vec3(0, 0, 0), vec3(0, 1, 0)), ray7(vec3(0, 0, 0), vec3(1, 0, 0)), ray7(vec3(1, 2, 3), vec3(0, -1, 0))])
    rgb, nseg, _, _ = oracle.probe_paths(w, rays, [1, 2, 3])
    assert list(nseg) == [1, 1, 1]
    assert np.allclose(rgb[0], (1, 1, 1), atol=1e-12) and np.allclose(rgb[1], (0.75, 0.85, 1), atol=1e-12)
    # depth = 0 hit on a lambertian -> (0,0,0) after one segment
    w2 = world_of(_dome(), r.hitable.sphere(center=vec3(0, 0, -5), radius=1.0, material=MATERIAL))
    rgb, nseg, _, _ = oracle.probe_paths(w2, ray7(vec3(0, 0, 0), vec3(0, 0, -1)), [7], depth=0)
    assert list(rgb[0]) == [0, 0, 0] and nseg[0] == 1
    # a miss returns accum = black (core.clj:40-41)
    w3 = world_of(r.hitable.sphere(center=vec3(0, 0, -5), radius=1.0, material=MATERIAL))
    rgb, nseg, _, _ = oracle.probe_paths(w3, ray7(vec3(0, 0, 0), vec3(0, 0, 1)), [7])
    assert list(rgb[0]) == [0, 0, 0] and nseg[0] == 1
    # metal whose fuzzed direction ends below the surface absorbs the path (shader.clj:56): fuzz 10 makes that common
    metal = r.shader.metal(albedo=r.texture.constant(color=vec3(1, 1, 1)), fuzz=10)
    w4 = world_of(_dome(), r.hitable.sphere(center=vec3(0, 0, -5), radius=1.0, material=metal))
    rays = np.tile(ray7(vec3(0, 0, 0), vec3(0, 0, -1)), (64, 1))
    rgb, nseg, _, _ = oracle.probe_paths(w4, rays, np.arange(64))
    absorbed = (rgb.sum(axis=1) == 0)
    assert absorbed.any() and (~absorbed).any()
    # total-rays <= 51 per sample with depth 50: two facing mirrors
    mirror = r.shader.metal(albedo=r.texture.constant(color=vec3(1, 1, 1)), fuzz=0.0)
    w5 = world_of(r.hitable.sphere(center=vec3(0, 0, -1001), radius=1000.0, material=mirror),
                  r.hitable.sphere(center=vec3(0, 0, 1001), radius=1000.0, material=mirror))
    rgb, nseg, _, _ = oracle.probe_paths(w5, ray7(vec3(0, 0, 0), vec3(0, 0, 1)), [9])
    assert nseg[0] == 51 and list(rgb[0]) == [0, 0, 0]


# ---- K14: quantiser (core.clj:52-57) ----------------------------------------------------------------------
def test_k14_quantiser(oracle):
    assert list(oracle.quantise([1.0, 0.25, 4.0])) == [255, 127, 255]
    assert list(oracle.quantise([float("nan"), 0.0, 1e-9])) == [0, 0, 0]


# ---- K15: furnace: closed lambertian albedo rho inside an emitter of radiance 1 -> sum rho^k --------------------
def test_k15_furnace(oracle):
    rho = 0.5
    lam = r.shader.lambertian(albedo=r.texture.constant(color=vec3(rho, rho, rho)))
    light = r.shader.diffuse_light(tex=r.texture.constant(color=vec3(1, 1, 1)))
    # a diffuse ball seen from outside, lit by a uniform dome: radiance of a convex lambertian = rho exactly
    w = world_of(r.hitable.sphere(center=vec3(0, 0, 0), radius=1000.0, material=light),
                 r.hitable.sphere(center=vec3(0, 0, -5), radius=1.0, material=lam))
    rays = np.tile(ray7(vec3(0, 0, 0), vec3(0, 0, -1)), (4096, 1))
    rgb, nseg, _, _ = oracle.probe_paths(w, rays, np.arange(4096) + 100)
    assert np.allclose(rgb, rho) and (nseg == 2).all()  # convex + uniform light: every path is exactly rho * 1


# ---- counter stream ---------------------------------------------------------------------------------------
def test_rng_matches_python_restatement(oracle):
    from raytrace_clj_amd.util import draw_bits, sample_key
    for seed, pix, s in [(0, 0, 0), (0x5EED0002, 12345, 63), ((1 << 64) - 1, (1 << 40) + 3, 4095)]:
        k = sample_key(seed, pix, s)
        assert oracle.sample_key(seed, pix, s) == k
        for d in (0, 1, 2, 1000):
            z = draw_bits(k, d)
            assert oracle.draw_bits(k, d) == z
            assert oracle.draw(k, d) == (z >> 11) * 2.0 ** -53
    # splitmix64 reference vector: seed 1234567 -> first outputs (public test vector of the generator)
    from raytrace_clj_amd.util import SplitMix64
    g = SplitMix64(1234567)
    assert [g.next_u64() for _ in range(3)] == [6457827717110365317, 3203168211198807973, 9817491932198370423]


def test_stream_mixer_avalanche():
    """the render stream's mixer (two rounds of x ^= x >> 32; x *= 0xD6E8FEB86659FD93, then x ^= x >> 32 -- it replaced the splitmix64 finaliser in
    round 3 because shifts by 32 are free on a 32-bit ALU): flipping any input bit flips every output bit with probability 1/2 to within the
    sampling noise, on random inputs and on the Weyl sequence key + GOLD * d the stream feeds it -- the same figures as the finaliser's
    (scripts/rng_quality.py prints both, and the one-multiply mixer that was rejected: rms 0.25)"""
    def mix(z):
        z = z.copy()
        for _ in range(2):
            z ^= z >> np.uint64(32)
            z *= np.uint64(0xD6E8FEB86659FD93)
        z ^= z >> np.uint64(32)
        return z
    from raytrace_clj_amd.util import mix64
    n = 40000
    rng = np.random.default_rng(3)
    with np.errstate(over="ignore"):
        assert [int(v) for v in mix(np.array([0, 1, 0x5EED0002, 2 ** 64 - 1], np.uint64))] == [mix64(0), mix64(1), mix64(0x5EED0002), mix64(2 ** 64 - 1)]
        for x in (rng.integers(0, 2 ** 64, n, dtype=np.uint64), np.uint64(0x1234567) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)):
            fx = mix(x)
            dev = np.zeros((64, 64))
            for i in range(64):
                d = fx ^ mix(x ^ np.uint64(1 << i))
                dev[i] = [(((d >> np.uint64(j)) & np.uint64(1)).mean() - 0.5) for j in range(64)]
            noise = 0.5 / np.sqrt(n)
            assert np.sqrt((dev ** 2).mean()) < 1.15 * noise and np.abs(dev).max() < 5.5 * noise


def test_rng_uniformity(oracle):
    k = oracle.sample_key(1, 2, 3)
    x = np.array([oracle.draw(k, d) for d in range(20000)])
    assert 0 <= x.min() and x.max() < 1 and abs(x.mean() - 0.5) < 0.01 and abs(x.var() - 1 / 12) < 0.005


# ---- BVH (hitable.clj:97-123) gives the same closest hit as the Hitlist scan --------------------
def test_bvh_equals_flat_scan(oracle, cover11_moving):
    w = fl.flatten(cover11_moving)
    rng = np.random.default_rng(5)
    n = 3000
    o = np.tile(vec3(13, 2, 3), (n, 1)) + rng.normal(0, 0.5, (n, 3))
    d = -o + rng.normal(0, 3.0, (n, 3))
    rays = np.concatenate([o, d, rng.random((n, 1))], axis=1)
    flat = oracle.probe_hit(w, rays)
    for seed in (1, 2):
        assert np.array_equal(flat, oracle.probe_hit(w, rays, bvh_seed=seed))


# ---- pixel/render plumbing (core.clj:43-57, 100-108) --------------------------------------------------------
def test_render_region_and_threads(oracle, cover_small):
    w = fl.flatten(cover_small)
    full, q, cnt = oracle.render(w, 32, 16, 3, depth=50, seed=77)
    assert full.shape == (16, 32, 3) and cnt[1] == 512 and 512 * 3 <= cnt[0] <= 512 * 3 * 51
    part, qp, _ = oracle.render(w, 32, 16, 3, depth=50, seed=77, region=(8, 4, 24, 12))
    assert np.array_equal(part, full[4:12, 8:24]) and np.array_equal(qp, q[4:12, 8:24])
    mt, qm, cm = oracle.render(w, 32, 16, 3, depth=50, seed=77, nthreads=4)
    assert np.array_equal(mt, full) and np.array_equal(qm, q) and cm[0] == cnt[0]
    other, _, _ = oracle.render(w, 32, 16, 3, depth=50, seed=78)
    assert not np.array_equal(other, full)
    # row 0 is the TOP of the picture: sky (bright, blue-ish) above ground
    assert full[0].mean() > 0.3


# ---- section 8(f3) records: no test in the reference covers them (parity unpinned by the reference); analytic KATs -------------
def nested(world):
    from oracle.tree import attach_tree
    f = fl.flatten(world, r.camera.PinholeCamera(*(np.zeros(3),) * 4))
    return attach_tree(f, world)


def test_f3_rectangles(oracle):
    H = r.hitable
    for ctor, o, d, p, n, uv in [
        (lambda: H.rect_xy(x0=-1, y0=-2, x1=3, y1=2, k=5, material=MATERIAL), vec3(1, 0, 0), vec3(0, 0, 1), (1, 0, 5), (0, 0, 1), (0.5, 0.5)),
        (lambda: H.rect_xz(x0=-1, z0=-2, x1=3, z1=2, k=5, material=MATERIAL), vec3(1, 0, 1), vec3(0, 2, 0), (1, 5, 1), (0, 1, 0), (0.5, 0.75)),
        (lambda: H.rect_yz(y0=-1, z0=-2, y1=3, z1=2, k=5, material=MATERIAL), vec3(0, 2, -1), vec3(1, 0, 0), (5, 2, -1), (1, 0, 0), (0.75, 0.25))]:
        w = nested(H.hitlist(items=[ctor()]))
        h = oracle.probe_hit(w, ray7(o, d), 0.001, FLT_MAX)[0]
        assert h[0] == 1 and tuple(h[3:6]) == p and tuple(h[6:9]) == n and tuple(h[9:11]) == uv
        flipped = nested(H.hitlist(items=[H.flip_normals(item=ctor())]))
        assert tuple(oracle.probe_hit(flipped, ray7(o, d), 0.001, FLT_MAX)[0, 6:9]) == tuple(-x for x in n)
        assert oracle.probe_hit(w, ray7(o, -d), 0.001, FLT_MAX)[0, 0] == 0  # behind the origin
    # inclusive interval and extents (hitable.clj:278,281-282): t == t-max and a hit exactly on the edge count
    w = nested(H.hitlist(items=[H.rect_xy(x0=0, y0=0, x1=1, y1=1, k=2, material=MATERIAL)]))
    assert oracle.probe_hit(w, ray7(vec3(1, 1, 0), vec3(0, 0, 1)), 0.0, 2.0)[0, 0] == 1
    assert oracle.probe_hit(w, ray7(vec3(1.0000001, 1, 0), vec3(0, 0, 1)), 0.0, 2.0)[0, 0] == 0
    # a rectangle tied with an EARLIER sphere wins (<=), a sphere tied with an earlier rectangle does not (<)
    s = H.sphere(center=vec3(0.5, 0.5, 3), radius=1.0, material=MATERIAL)
    rect = H.rect_xy(x0=0, y0=0, x1=1, y1=1, k=2, material=MATERIAL)
    ray = ray7(vec3(0.5, 0.5, 0), vec3(0, 0, 1))
    assert oracle.probe_hit(nested(H.hitlist(items=[s, rect])), ray, 0.001, FLT_MAX)[0, 1] == 1
    assert oracle.probe_hit(nested(H.hitlist(items=[rect, s])), ray, 0.001, FLT_MAX)[0, 1] == 0


def test_f3_triangle(oracle):
    H = r.hitable
    tri = H.triangle(v0=vec3(0, 0, 0), v1=vec3(0, 1, 0), v2=vec3(1, 0, 0), material=MATERIAL)  # as make-two-triangles (scene.clj:103-107)
    w = nested(H.hitlist(items=[tri]))
    h = oracle.probe_hit(w, ray7(vec3(0.25, 0.25, -10), vec3(0, 0, 1)), 0.001, FLT_MAX)[0]
    assert h[0] == 1 and h[2] == 10.0 and tuple(h[3:6]) == (0.25, 0.25, 0.0) and tuple(h[9:11]) == (0.25, 0.25)
    assert tuple(h[6:9]) == (0.0, 0.0, -1.0)  # cross(v0v1, v0v2), not normalised
    assert oracle.probe_hit(w, ray7(vec3(0.25, 0.25, 10), vec3(0, 0, -1)), 0.001, FLT_MAX)[0, 0] == 0  # one sided (det > 1e-8)
    assert oracle.probe_hit(w, ray7(vec3(0.75, 0.75, -10), vec3(0, 0, 1)), 0.001, FLT_MAX)[0, 0] == 0  # u + v > 1


def test_f3_instances_and_box(oracle):
    H = r.hitable
    b = H.box(p0=vec3(0, 0, 0), p1=vec3(2, 2, 2), material=MATERIAL)
    w = nested(H.hitlist(items=[b]))
    h = oracle.probe_hit(w, ray7(vec3(1, 1, -5), vec3(0, 0, 1)), 0.001, FLT_MAX)[0]
    assert h[0] == 1 and h[2] == 5.0 and tuple(h[6:9]) == (0.0, 0.0, -1.0)  # near face z = 0 is the flipped RectXY
    h = oracle.probe_hit(w, ray7(vec3(1, 1, 1), vec3(0, 0, 1)), 0.001, FLT_MAX)[0]
    assert h[2] == 1.0 and tuple(h[6:9]) == (0.0, 0.0, 1.0)
    moved = H.translate(item=b, offset=vec3(10, 0, 0))
    h = oracle.probe_hit(nested(H.hitlist(items=[moved])), ray7(vec3(11, 1, -5), vec3(0, 0, 1)), 0.001, FLT_MAX)[0]
    assert h[0] == 1 and h[2] == 5.0 and tuple(h[3:6]) == (11.0, 1.0, 0.0)
    rot = H.rotate_y(item=H.rect_xy(x0=-1, y0=-1, x1=1, y1=1, k=0, material=MATERIAL), theta=90.0)  # z = 0 plane -> x = 0 plane
    h = oracle.probe_hit(nested(H.hitlist(items=[rot])), ray7(vec3(-5, 0.5, 0.25), vec3(1, 0, 0)), 0.001, FLT_MAX)[0]
    assert h[0] == 1 and abs(h[2] - 5.0) < 1e-12 and np.allclose(h[3:6], (0, 0.5, 0.25), atol=1e-12) and np.allclose(np.abs(h[6:9]), (1, 0, 0), atol=1e-12)
    # host mirror bboxes (hitable.clj:292-294, 397-400, 452-481, 572-577)
    assert np.allclose(moved.bbox(0, 1).vmin, (10, 0, 0)) and np.allclose(moved.bbox(0, 1).vmax, (12, 2, 2))
    rb = rot.bbox(0, 1)
    assert np.allclose(rb.vmin, (-1e-4, -1, -1), atol=1e-9) and np.allclose(rb.vmax, (1e-4, 1, 1), atol=1e-9)


def test_f3_flatten_chains():
    f = fl.flatten(r.scene.make_cornell_box(64, 64))
    assert f.n_prims == 18 and (f.prim_kind >= 3).all() and f.prim_flip.sum() == 9
    assert len(f.xform_kind) == 4 and list(f.xform_kind) == [0, 1, 0, 1] and (f.prim_xform[:, 1] == 2).sum() == 12
    foggy = fl.flatten(r.scene.make_cornell_box(64, 64, classic=False))
    assert foggy.n_prims == 8 and (foggy.prim_kind[:8] == 7).sum() == 2 and (foggy.prim_kind[8:] & 16).all() and len(foggy.prim_kind) == 20
    # the same Box instanced twice is two sets of primitives
    b = r.hitable.box(p0=vec3(0, 0, 0), p1=vec3(1, 1, 1), material=MATERIAL)
    f2 = fl.flatten(r.hitable.hitlist(items=[r.hitable.translate(item=b, offset=vec3(5, 0, 0)), r.hitable.translate(item=b, offset=vec3(-5, 0, 0))]), None)
    assert f2.n_prims == 12 and len(f2.xform_kind) == 2


def test_f3_cornell_and_triangles_render(oracle):
    from oracle.tree import flatten_with_tree
    lin, q, cnt = oracle.render(flatten_with_tree(r.scene.make_cornell_box(32, 32)), 32, 32, 16, 50, 7, nthreads=8)
    # camera u = vup x w = (-1, 0, 0): the image's left is world +x, i.e. the green wall (x = 555); the red wall (x = 0) is on the right
    assert np.isfinite(lin).all() and lin.mean() > 0.02
    assert lin[:, :8, 1].mean() > lin[:, :8, 0].mean() and lin[:, -8:, 0].mean() > lin[:, -8:, 1].mean()
    lin2, _, _ = oracle.render(flatten_with_tree(r.scene.make_two_triangles(32, 16)), 32, 16, 8, 50, 7, nthreads=8)
    assert lin2[:, :, 2].mean() > lin2[:, :, 0].mean()  # blue-ish light dome dominates


# ---- section 8(f4) textures (texture.clj:60-138, perlin.clj): no reference test covers them; properties + restated values ----------
def _py_noise(vec, perm, p):
    """independent pure-Python restatement of perlin.clj:19-50"""
    import math
    ijk = [math.floor(x) for x in p]
    uvw = [p[k] - ijk[k] for k in range(3)]
    uu, vv, ww = [(t * t) * (3 - 2 * t) for t in uvw]
    acc = None
    for di in (0, 1):
        for dj in (0, 1):
            for dk in (0, 1):
                c = vec[perm[0][(ijk[0] + di) & 255] ^ perm[1][(ijk[1] + dj) & 255] ^ perm[2][(ijk[2] + dk) & 255]]
                wv = (uvw[0] - di, uvw[1] - dj, uvw[2] - dk)
                term = (((di * uu + (1.0 - di) * (1.0 - uu)) * (dj * vv + (1.0 - dj) * (1.0 - vv))) * (dk * ww + (1.0 - dk) * (1.0 - ww))) * \
                       ((wv[0] * c[0] + wv[1] * c[1]) + wv[2] * c[2])
                acc = term if acc is None else acc + term
    return acc


def test_f4_perlin_tables_and_noise(oracle):
    from raytrace_clj_amd import perlin
    vec, perm = perlin.make_tables()
    assert np.allclose(np.linalg.norm(vec, axis=1), 1.0, atol=1e-15) and all(sorted(perm[a]) == list(range(256)) for a in range(3))
    v2, p2 = perlin.make_tables()
    assert np.array_equal(vec, v2) and np.array_equal(perm, p2) and not np.array_equal(perlin.make_tables(7)[0], vec)
    T = r.texture
    world = r.hitable.hitlist(items=[r.hitable.sphere(center=vec3(0, 0, 0), radius=1.0, material=r.shader.lambertian(albedo=t)) for t in
                                     (T.perlin_noise(scale=1.0), T.perlin_turbulence(scale=4, depth=7), T.marble(scale=4, depth=5))])
    f = fl.flatten(world, r.camera.PinholeCamera(*(np.zeros(3),) * 4))
    assert list(f.tex_kind) == [3, 4, 5] and f.perlin_vectors.shape == (256, 3)
    rng = np.random.default_rng(0)
    pts = rng.normal(0, 7, (300, 3))
    uvp = np.concatenate([np.zeros((300, 2)), pts], axis=1)
    got = oracle.probe_texture(f, 0, uvp)
    exp = np.array([0.5 * (_py_noise(vec, perm, p) + 1.0) for p in pts])
    assert np.array_equal(got[:, 0], exp) and np.array_equal(got[:, 0], got[:, 2])
    assert np.abs(2 * got[:, 0] - 1).max() < 1.0 and np.abs(2 * got[:, 0] - 1).max() > 0.2
    # noise vanishes on the integer lattice (every weight vector with non-zero hermite weight is zero there)
    lat = np.concatenate([np.zeros((27, 2)), np.array([[i, j, k] for i in (-2, 0, 5) for j in (-1, 0, 3) for k in (0, 1, -7)], float)], axis=1)
    assert np.array_equal(oracle.probe_texture(f, 0, lat), np.full((27, 3), 0.5))
    turb = oracle.probe_texture(f, 1, uvp)[:, 0]
    assert (turb >= 0.5).all() and turb.max() < 2.0  # 0.5 * (|sum| + 1)
    marble = oracle.probe_texture(f, 2, uvp)[:, 0]
    assert (marble >= 0).all() and (marble <= 1).all()


def test_f4_flip_and_image(oracle):
    T = r.texture
    img = np.zeros((2, 4, 3), np.uint8)
    img[0, :, 0] = [10, 20, 30, 40]; img[1, :, 1] = [50, 60, 70, 80]
    grad = T.uv_gradient(co=vec3(1, 0, 0), cu=vec3(0, 1, 0), cv=vec3(0, 0, 1), cuv=vec3(1, 1, 1))
    texs = [T.image_map(image=img), T.flip_texture_v(tex=T.image_map(image=img)), T.flip_texture_u(tex=grad), grad,
            T.flip_texture_u(tex=T.flip_texture_v(tex=T.image_map(image=img)))]
    world = r.hitable.hitlist(items=[r.hitable.sphere(center=vec3(0, 0, 0), radius=1.0, material=r.shader.lambertian(albedo=t)) for t in texs])
    f = fl.flatten(world, r.camera.PinholeCamera(*(np.zeros(3),) * 4))
    def sample(mat, u, v):
        return oracle.probe_texture(f, int(f.mat_tex[mat]), [u, v, 0, 0, 0])[0]
    assert np.allclose(sample(0, 0.30, 0.10), (20 / 255.0, 0, 0)) and np.allclose(sample(0, 0.99, 0.75), (0, 80 / 255.0, 0))
    assert np.allclose(sample(1, 0.30, 0.90), (20 / 255.0, 0, 0))          # FlipTextureV: v -> 1 - v
    assert np.allclose(sample(4, 0.70, 0.90), (20 / 255.0, 0, 0))          # FlipTextureU(FlipTextureV(image))
    assert np.allclose(sample(2, 0.25, 0.6), sample(3, 0.75, 0.6))         # FlipTextureU: u -> 1 - u
    assert np.allclose(sample(0, 1.0, 1.0), (0, 80 / 255.0, 0))            # u = 1.0: clamped (the reference indexes out of bounds)


def test_f4_scenes_render(oracle):
    from oracle.tree import flatten_with_tree
    for sc, (nx, ny) in [(r.scene.make_two_perlin_spheres(32, 16), (32, 16)), (r.scene.make_textured_sphere(32, 16), (32, 16)),
                         (r.scene.make_example_light(32, 16), (32, 16))]:
        lin, q, cnt = oracle.render(flatten_with_tree(sc), nx, ny, 4, 50, 3, nthreads=8)
        assert np.isfinite(lin).all() and lin.mean() > 0.05 and cnt[1] == nx * ny


# ---- ConstantMedium + Isotropic (hitable.clj:516-546, shader.clj:129-143): no reference test; analytic properties ------------------
def test_medium_semantics(oracle):
    from oracle.tree import attach_tree
    H, S, T = r.hitable, r.shader, r.texture
    ball = H.sphere(center=vec3(0, 0, -5), radius=1.0, material=S.dielectric(ri=1.5))
    light = H.sphere(center=vec3(0, 0, 0), radius=100.0, material=S.diffuse_light(tex=T.constant(color=vec3(1, 1, 1))))
    def world(density, albedo=(0.5, 0.25, 1.0)):
        w = H.make_bvh([light, H.constant_medium(boundary=ball, density=density, albedo=T.constant(color=vec3(*albedo)))], 0.0, 1.0)
        f = fl.flatten(w, r.camera.PinholeCamera(*(np.zeros(3),) * 4))
        return attach_tree(f, w), f
    ray = ray7(vec3(0, 0, 0), vec3(0, 0, -1), 0.25)
    # flattened form: the medium is a world primitive, its boundary sphere follows the world flagged PRIM_BOUNDARY
    ft, f = world(1e9)
    assert f.n_prims == 2 and len(f.prim_kind) == 3 and sorted(f.prim_kind[:2]) == [0, 7] and f.prim_kind[2] == 16
    m = int(np.flatnonzero(f.prim_kind == 7)[0])
    assert f.prim_geom[m, 0] == 1e9 and f.prim_geom[m, 1] == 2 and f.prim_geom[m, 2] == 1 and f.mat_kind[f.prim_mat[m]] == 4
    # density -> infinity: scatters at the entry point t = 4 (hit-distance -> 0); normal (1,0,0), uv (0,0) are arbitrary constants
    h = oracle.probe_hit(ft, ray)[0]
    assert h[0] == 1 and h[1] == m and abs(h[2] - 4.0) < 1e-6 and tuple(h[6:9]) == (1.0, 0.0, 0.0)
    # density -> 0: never scatters inside; only the light behind is hit
    assert oracle.probe_hit(world(1e-12)[0], ray)[0, 1] != m
    # a ray that misses the boundary, or whose segment lies behind the origin, draws nothing and hits nothing of the medium
    assert oracle.probe_hit(ft, ray7(vec3(0, 3, 0), vec3(0, 0, -1)))[0, 1] != m
    assert oracle.probe_hit(ft, ray7(vec3(0, 0, -10), vec3(0, 0, -1)))[0, 1] != m
    # from inside the boundary the segment starts at t-min
    hin = oracle.probe_hit(ft, ray7(vec3(0, 0, -5), vec3(0, 0, -1)))[0]
    assert hin[1] == m and 0.001 <= hin[2] < 0.0011
    # Isotropic.scatter: direction = a point of the unit ball, attenuation = albedo, and the new ray's TIME is the hit's t (shader.clj:136)
    sc_out = oracle.probe_scatter(f, int(f.prim_mat[m]), ray, np.array([0, 0, -4, 1, 0, 0, 0, 0], float), [77])[0]
    assert sc_out[0] == 1 and np.linalg.norm(sc_out[1:4]) < 1 and tuple(sc_out[4:7]) == (0.5, 0.25, 1.0) and sc_out[8] >= 3
    # mean free path: P(scatter inside a chord of length 2) = 1 - exp(-2 density)
    ft2, _ = world(0.7)
    rays = np.tile(ray, (20000, 1))
    rgb, nseg, log, nlog = oracle.probe_paths(ft2, rays, np.arange(20000) + 5, depth=50, max_seg=1)
    frac = (log[:, 0, 0] == m).mean()
    assert abs(frac - (1 - np.exp(-1.4))) < 0.01
    # a medium under a Hitlist: its t-max comes narrowed by the items before it -- flagged for the device (RTMI_MEDIA_HITLIST, round 3) ...
    fh = fl.flatten(H.hitlist(items=[light, H.constant_medium(boundary=ball, density=1.0, albedo=T.constant(color=vec3(1, 1, 1)))]), None)
    assert fh.media_mode == 1 and list(fh.media_calls) == [1]
    # ... and the nested oracle narrows it: a wall at t = 3, listed BEFORE a thick fog whose ball spans t in [2, 6] on this ray, clips the fog's
    # segment to [2, 3]; listed AFTER the fog it does not (hitable.clj:15-26 hands every item the closest hit SO FAR)
    from oracle.tree import flatten_with_tree
    wall = H.rect_xy(x0=-5, y0=-5, x1=5, y1=5, k=-3.0, material=S.lambertian(albedo=T.constant(color=vec3(0.5, 0.5, 0.5))))
    fogball = H.sphere(center=vec3(0, 0, -4), radius=2.0, material=S.dielectric(ri=1.5))
    def listed(items):
        cam = r.camera.pinhole_camera(lookfrom=vec3(0, 0, 0), lookat=vec3(0, 0, -1), vup=vec3(0, 1, 0), vfov=40.0, aspect=1.0)
        return flatten_with_tree({"camera": cam, "world": H.hitlist(items=items)})
    thin = 0.05  # the free path (mean 20) mostly exceeds a 1-long segment and mostly not a 4-long one... the draws are the same in both scenes
    n = 4000
    rays = np.tile(ray7(vec3(0, 0, 0), vec3(0, 0, -1)), (n, 1))
    keys = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    hits = {}
    for name, items in (("wall first", [wall, H.constant_medium(boundary=fogball, density=thin * 10, albedo=T.constant(color=vec3(1, 1, 1)))]),
                        ("fog first", [H.constant_medium(boundary=fogball, density=thin * 10, albedo=T.constant(color=vec3(1, 1, 1))), wall])):
        ft = listed(items)
        mi = int(np.flatnonzero((ft.prim_kind[:ft.n_prims] & 15) == 7)[0])
        rgb, nseg, log, nlog = oracle.probe_paths(ft, rays, keys, depth=1, ctr0=0, max_seg=1)
        t = log[:, 0, 1]
        hits[name] = (log[:, 0, 0] == mi, t)
        assert t.max() <= 3.0 + 1e-12, name  # nothing is ever reported behind the wall
    (in_a, t_a), (in_b, t_b) = hits["wall first"], hits["fog first"]
    # the same draw scatters inside the fog in both scenes iff its free path fits the CLIPPED segment [2, 3]; with the fog listed first the
    # un-clipped segment [2, 6] accepts more draws, but those beyond t = 3 lose to the wall listed after it
    assert abs(in_a.mean() - (1 - np.exp(-0.5 * 1.0))) < 0.03 and np.all(t_a[in_a] <= 3.0) and np.all(t_a[in_a] >= 2.0)
    assert np.array_equal(in_a, in_b) and np.allclose(t_a, t_b), "for this geometry both orders agree hit for hit (the wall clips either way)"


def test_media_scenes_render(oracle):
    from oracle.tree import flatten_with_tree
    for sc, (nx, ny) in [(r.scene.make_subsurface_sphere(32, 16), (32, 16)), (r.scene.make_cornell_box(24, 24, classic=False), (24, 24)),
                         (r.scene.make_final(24, 24), (24, 24))]:
        lin, q, cnt = oracle.render(flatten_with_tree(sc), nx, ny, 4, 50, 3, nthreads=8)
        assert np.isfinite(lin).all() and lin.mean() > 0.05 and cnt[1] == nx * ny

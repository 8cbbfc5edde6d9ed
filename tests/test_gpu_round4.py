"""GPU tests added in round 4: the helpers the mixed-kind kernels gained (the refined reciprocal of a signed divisor behind a rectangle's t, the
medium's own log of a draw, the explicit slot count of the math probe), the scan of small mixed-kind worlds against the culled scan and the tree,
Box runs as one leaf, and the bench line's new keys.  Everything goes through the C-ABI."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import raytrace_clj_amd as r
from raytrace_clj_amd import core
from raytrace_clj_amd import flatten as fl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_signed_refined_reciprocal_division():
    """div_by (a rectangle's t = (k - o_a) / d_a with the local direction's refined reciprocal, hitable.clj:269-363): the IEEE quotient bit for bit for
    divisors of EITHER sign wherever the division would not scale its operands; out-of-range divisors take the plain division per lane; numerators
    the division would scale give a value that passes / fails t-min <= t <= closest like the quotient."""
    rng = np.random.default_rng(11)
    n = 64 * 400
    a = (np.abs(rng.normal(0, 1, n)) * 10.0 ** rng.integers(-6, 7, n) + 1e-12) * rng.choice([-1.0, 1.0], n)
    num = rng.normal(0, 1, n) * 10.0 ** rng.integers(-12, 13, n)
    o = core.probe_math(np.stack([num, a, np.zeros(n)], axis=1))
    assert np.all(o[:, 11] == 1.0) and np.array_equal(o[:, 10], num / a)
    q = rng.uniform(1.0, 2.0, n)  # hard-to-round quotients around exact products
    hard = (q * a) * (1.0 + rng.integers(-3, 4, n) * 2.0 ** -53)
    o = core.probe_math(np.stack([hard, a, np.zeros(n)], axis=1))
    assert np.all(o[:, 11] == 1.0) and np.array_equal(o[:, 10], hard / a)
    a2 = a.copy()  # per lane: out-of-range divisors (an axis-parallel ray's zero component among them) take the plain division
    a2[::64] = np.resize([1e-80, -1e80, 0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324], len(a2[::64]))
    o = core.probe_math(np.stack([num, a2, np.zeros(n)], axis=1))
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        ref = num / a2
    slow = np.zeros(n, bool)
    slow[::64] = True
    assert np.all(o[slow, 11] == 0.0) and np.all(o[~slow, 11] == 1.0) and np.array_equal(o[:, 10], ref, equal_nan=True)
    for tmin, tmax in ((0.0, 3.4e38), (-3.4028234663852886e38, 3.4028234663852886e38), (0.001, np.inf)):  # a medium's boundary scans run from -FMAX: plain division
        o = core.probe_math(np.stack([num, a, np.zeros(n)], axis=1), tmin=tmin, tmax=tmax)
        assert np.all(o[:, 11] == 0.0) and np.array_equal(o[:, 10], num / a)
    ext = np.resize(np.array([0.0, -0.0, 5e-324, -1e-310, 1e-300, -1e-295, 1e250, -1e290, 1.7e308, np.inf, -np.inf, np.nan]), n)
    tmin, tmax = 0.001, 3.4028234663852886e38
    o = core.probe_math(np.stack([ext, a, np.zeros(n)], axis=1), tmin=tmin, tmax=tmax)
    with np.errstate(invalid="ignore", over="ignore", under="ignore"):
        ref = ext / a
    keep = (ref >= tmin) & (ref <= tmax)  # a rectangle's inclusive interval
    assert np.array_equal((o[:, 10] >= tmin) & (o[:, 10] <= tmax), keep) and np.array_equal(o[keep, 10], ref[keep])


def test_medium_log_of_a_draw():
    """rt_log_unit: log of a draw in [0, 1) (hitable.clj:529) within one ulp of the host libm's on every kind of draw the stream produces -- k 2^-53 for
    53-bit k, values next to 1, the smallest draws, powers of two and their neighbours -- and -inf at 0 (no hit)."""
    rng = np.random.default_rng(5)
    k = np.concatenate([rng.integers(1, 2 ** 53, 400000), 2 ** 53 - rng.integers(1, 4096, 50000), rng.integers(1, 4096, 50000),
                        2 ** rng.integers(0, 53, 2000), 2 ** rng.integers(1, 53, 2000) - 1, 2 ** rng.integers(1, 52, 2000) + 1]).astype(np.float64)
    u = k * 2.0 ** -53
    o = core.probe_math(np.stack([u, np.ones_like(u), np.zeros_like(u)], axis=1))[:, 9]
    ref = np.log(u)
    err = np.abs(o - ref) / np.spacing(np.abs(ref))
    assert err.max() <= 1.0, err.max()
    o = core.probe_math(np.array([[0.0, 1.0, 0.0], [1.0, 1.0, 0.0], [0.5, 1.0, 0.0]]))[:, 9]
    assert o[0] == -np.inf and o[1] == 0.0 and o[2] == np.log(0.5)


def test_math_probe_slot_count_is_the_callers():
    """rtmi_probe_math2 writes exactly n_slots values per triple; rtmi_probe_math (the entry as first published) writes EIGHT -- a host built against
    the first header is not overrun (library version 203 wrote nine into that signature)."""
    L = r._ffi.lib()
    assert L.rtmi_version() >= 204
    ctx = core.default_context()
    abc = np.ascontiguousarray(np.array([[0.25, 2.0, 1.5], [0.5, -3.0, 2.5]]))
    out = np.full(2 * 8 + 4, -777.0)
    core.check(L.rtmi_probe_math(ctx.handle, 2, core.ptr(abc), 0.001, 3.4e38, core.ptr(out)))
    assert np.all(out[16:] == -777.0) and out[0] == 0.5 and out[8] == np.sqrt(0.5)
    for slots in (1, 9, 12):
        out = np.full(2 * slots + 3, -777.0)
        core.check(L.rtmi_probe_math2(ctx.handle, 2, core.ptr(abc), 0.001, 3.4e38, slots, core.ptr(out)))
        assert np.all(out[2 * slots:] == -777.0) and out[0] == 0.5 and out[slots] == np.sqrt(0.5)
    assert L.rtmi_probe_math2(ctx.handle, 2, core.ptr(abc), 0.001, 3.4e38, 13, core.ptr(out)) != 0
    assert L.rtmi_probe_math2(ctx.handle, 2, core.ptr(abc), 0.001, 3.4e38, 0, core.ptr(out)) != 0


def _device_scene(flat, accel):
    ctx = core.default_context()
    ctx.set_option("accel", accel)
    return core.DeviceScene(flat), ctx


def _probe_rays(flat, n, seed, accel):
    """closest hits of n rays shot through the scene's bounding region, by the flat scan (0) or the tree (1)"""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-50, 600, (n, 3))
    d = rng.normal(0, 1, (n, 3))
    rays = np.concatenate([o, d, rng.uniform(0, 1, (n, 1))], axis=1)
    ds, ctx = _device_scene(flat, accel)
    try:
        return ds.probe_hit(rays, 0.001, 3.4028234663852886e38)
    finally:
        ds.close()
        ctx.set_option("accel", 1)


def _frame(flat, accel, nx, ns):
    ds, ctx = _device_scene(flat, accel)
    try:
        return ds.render(nx, nx, ns)
    finally:
        ds.close()
        ctx.set_option("accel", 1)


def _cornell_smoke(nx, ny):
    return r.scene.make_cornell_box(nx, ny, classic=False)


@pytest.mark.parametrize("make", [r.scene.make_cornell_box, _cornell_smoke, r.scene.make_two_triangles, r.scene.make_example_light, r.scene.make_subsurface_sphere])
def test_small_world_scan_is_bit_identical_to_the_culled_scan_and_the_tree(make, monkeypatch):
    """scan_small_ext (the scan of a world of at most RTMI_SMALL_SCAN_MAX primitives: shared local rays, refined reciprocals, no cull) against the culled
    scan (RTMI_SMALL_SCAN=0 at scene creation) and the tree: hit records of 60 000 rays bit for bit, and whole frames with their counters."""
    flat = fl.flatten(make(96, 96))
    monkeypatch.setenv("RTMI_FLAT_BELOW", "0")
    ref = _probe_rays(flat, 60000, 3, 1)          # the tree
    monkeypatch.setenv("RTMI_SMALL_SCAN", "0")
    culled = _probe_rays(flat, 60000, 3, 0)       # the culled scan
    f_culled = _frame(flat, 0, 96, 16)
    monkeypatch.delenv("RTMI_SMALL_SCAN")
    small = _probe_rays(flat, 60000, 3, 0)        # the small-world scan
    assert np.array_equal(culled, ref, equal_nan=True) and np.array_equal(small, ref, equal_nan=True)
    assert (ref[:, 0] > 0).sum() > 1000
    f_tree, f_small = _frame(flat, 1, 96, 16), _frame(flat, 0, 96, 16)
    for lin, q, cnt in (f_culled, f_small):
        assert np.array_equal(lin, f_tree[0]) and np.array_equal(cnt, f_tree[2])


def test_box_runs_as_one_leaf_are_bit_identical(monkeypatch):
    """RTMI_BOX_LEAF=1: six consecutive rectangles that form a Box (hitable.clj:500-511) become one leaf of the tree (ext_box_test: one local ray, three
    refined reciprocals, six faces under their own indices).  Same hits, same frames, same counters as six leaves -- make-final's 400 ground boxes and
    the Cornell box's two rotated ones."""
    for make, nx, ns in ((r.scene.make_final, 64, 8), (r.scene.make_cornell_box, 64, 8)):
        flat = fl.flatten(make(nx, nx))
        monkeypatch.setenv("RTMI_FLAT_BELOW", "0")
        out = []
        for box in ("0", "1"):
            monkeypatch.setenv("RTMI_BOX_LEAF", box)
            hits = _probe_rays(flat, 40000, 9, 1)
            lin, q, cnt = _frame(flat, 1, nx, ns)
            out.append((hits, lin, cnt))
        assert np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def test_unread_uv_coordinate_moves_a_sky_colour_by_a_few_ulps_at_most(oracle, cover11):
    """The one deliberate device-vs-oracle deviation: a UVGradient whose corner colours do not vary along u (the cover scene's
    sky, scene.clj:340-344: co = cu, cv = cuv, compared bit for bit at scene creation) is evaluated with u = 1/2 on the device -- no atan2 --, with the
    reference's own u in the oracle: c (1 - u) + c u against c.  Pinned here: on 20 000 rays that end on the dome the two colours differ by a few ulps per
    channel at most (bound: 4), and nothing else differs -- same primitive, same t, p, normal bit for bit, same segment count (the dome scatters nothing: no draws)."""
    flat = fl.flatten(cover11)
    rng = np.random.default_rng(12)
    n = 20000
    d = rng.normal(0, 1, (n, 3))
    d[:, 1] = np.abs(d[:, 1]) + 0.05                       # upwards: the sky
    o = np.tile([0.0, 30.0, 0.0], (n, 1)) + rng.uniform(-5, 5, (n, 3))  # above every sphere
    rays = np.concatenate([o, d, rng.uniform(0, 1, (n, 1))], axis=1)
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    ergb, enseg, elog, enlog = oracle.probe_paths(flat, rays, keys, depth=50, ctr0=0, max_seg=2)
    ds = core.DeviceScene(flat)
    rgb, nseg, log, nlog = ds.probe_paths(rays, keys, depth=50, ctr0=0, max_seg=2)
    ds.close()
    dome = int(np.flatnonzero((flat.prim_kind == fl.PRIM_UVSPHERE) & (flat.prim_geom[:, 3] == 1000.0))[0])  # (the flattener lists the world in bvh-node order)
    assert np.all(enseg == 1) and np.array_equal(nseg, enseg) and np.all(elog[:, 0, 0] == dome)   # one segment, ending on the dome
    assert np.array_equal(log, elog)                                                         # index, t, p, normal: bit for bit
    # c (1 - u) + c u is c within one ulp for each of the two u-lerps; the v-lerp of the two carries them on, and v itself (asin: the path's own within 2 ulp
    # of the host libm's, test_table_atan2_asin_against_libm) moves the colour by |dv| |a - b| <= an ulp: four ulps bound the sum, most colours are equal outright
    ulps = np.abs(rgb - ergb) / np.spacing(np.abs(ergb))
    assert ulps.max() <= 4.0 and np.abs(ergb).min() > 0.4, ulps.max()
    assert (rgb != ergb).mean() < 0.5 and ulps.mean() < 0.5


def test_a_medium_listed_twice_in_a_hitlist_matches_the_nested_oracle(oracle):
    """Hitlist.hit? asks every item in turn with the closest hit so far as t-max (hitable.clj:15-26): a ConstantMedium record that stands in the list TWICE is
    asked twice -- two draws, the second narrowed by everything before the second listing (round 3 rejected such worlds).  The flatteners now make every listing
    a primitive of its own at its own place (same boundary, same phase function); the device's RTMI_MEDIA_HITLIST scan then needs nothing new.  Checked against
    the oracle's nested evaluation: segment logs, frames, BVH = flat scan; and the second listing must matter (the frame differs from the once-listed world's)."""
    from oracle.tree import flatten_with_tree
    from tests.test_gpu_round3 import hitlist_media_scene, rms, RMS_TOL
    sc = hitlist_media_scene(fog_twice=True)
    f = flatten_with_tree(sc)
    assert f.media_mode == 1 and len(f.media_calls) == 3 and list(f.media_calls) == sorted(set(int(m) for m in f.media_calls))
    a, b = int(f.media_calls[0]), int(f.media_calls[2])
    assert np.array_equal(f.prim_geom[a][:1], f.prim_geom[b][:1]) and f.prim_mat[a] == f.prim_mat[b]  # the same density and phase function, two places
    nx, ny, ns = 64, 32, 8
    exp_lin, exp_q, exp_cnt = oracle.render(f, nx, ny, ns, 50, 0x5EED0002, nthreads=16)
    rng = np.random.default_rng(2)
    n = 4096
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    cam = oracle.probe_camera(f, rng.random((n, 2)), keys)
    ctr0 = int(cam[:, 7].max())
    ergb, enseg, elog, enlog = oracle.probe_paths(f, cam[:, :7], keys, depth=50, ctr0=ctr0, max_seg=6)
    logged = np.arange(6)[None, :] < enlog[:, None]
    assert (elog[:, :, 0][logged].astype(int) == b).sum() > 5, "paths must scatter at the SECOND listing too"
    ctx = core.Context(0)
    ds = core.DeviceScene(f, ctx=ctx)
    frames = {}
    for accel in (1, 0):
        ctx.set_option("accel", accel)
        rgb, nseg, log, nlog = ds.probe_paths(cam[:, :7], keys, depth=50, ctr0=ctr0, max_seg=6)
        same = nseg == enseg
        assert same.mean() > 0.995, accel
        assert np.array_equal(log[same][:, :, 0], elog[same][:, :, 0]) and np.allclose(log[same], elog[same], rtol=1e-9, atol=1e-9), accel
        lin, q, cnt = ds.render(nx, ny, ns)
        assert abs(int(cnt[0]) - int(exp_cnt[0])) <= 52 and rms(lin, exp_lin) <= RMS_TOL, (accel, rms(lin, exp_lin), cnt, exp_cnt)
        frames[accel] = (lin, cnt)
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1])
    ds.close()
    once = core.DeviceScene(flatten_with_tree(hitlist_media_scene(fog_twice=False)), ctx=ctx)
    lin1, q1, cnt1 = once.render(nx, ny, ns)
    once.close(); ctx.close()
    assert int(cnt1[0]) != int(frames[1][1][0])


def narrowed_media_scene(seed=9):
    """Hitlists holding media BELOW bvh-nodes (round 3 rejected such worlds): the world is a bvh-node tree built by hand (the reference's Hitlist has no bbox, so
    make-bvh cannot sort one: its box is the union of its items' boxes here); one leaf is a Hitlist [surfaces, fog, surfaces, smoke, surfaces] -- the fog's t-max is
    narrowed by the surfaces before it IN THAT LIST only, the smoke's by those, the fog and the surfaces between --, another leaf a second Hitlist with a medium of
    its own, and a haze medium sits in the tree itself (un-narrowed, and asked twice: a one-item make-bvh)."""
    from raytrace_clj_amd.util import vec3
    rng = np.random.default_rng(seed)
    H, S, T = r.hitable, r.shader, r.texture
    grey = S.lambertian(albedo=T.constant(color=vec3(0.6, 0.6, 0.6)))

    def ball(cx, cz):
        return H.sphere(center=vec3(cx + rng.uniform(-1, 1), rng.uniform(0.3, 2.0), cz + rng.uniform(-1, 1)), radius=float(rng.uniform(0.25, 0.6)),
                        material=[grey, S.metal(albedo=T.constant(color=vec3(0.8, 0.7, 0.6)), fuzz=0.1), S.dielectric(ri=1.5)][int(rng.integers(0, 3))])

    def union(items):
        bs = [it.bbox(0.0, 1.0) for it in items]
        return H.AABB(np.min([b.vmin for b in bs], axis=0), np.max([b.vmax for b in bs], axis=0))

    class BoxedHitlist(H.Hitlist):  # a Hitlist that can tell its box (what a maintainer would add to put one below make-bvh)
        def bbox(self, t0, t1):
            return union(self.items)

    fog = H.constant_medium(boundary=H.sphere(center=vec3(-3, 1.5, 0), radius=2.2, material=S.dielectric(ri=1.5)), density=0.4, albedo=T.constant(color=vec3(0.9, 0.9, 0.9)))
    smoke = H.constant_medium(boundary=H.box(p0=vec3(-5.5, 0, 1.5), p1=vec3(-3.0, 2, 4), material=grey), density=0.9, albedo=T.constant(color=vec3(0.1, 0.1, 0.1)))
    list1 = BoxedHitlist(items=[ball(-3, 0) for _ in range(5)] + [fog] + [ball(-3, 1) for _ in range(4)] + [smoke] + [ball(-4, 2) for _ in range(3)])
    mist = H.constant_medium(boundary=H.sphere(center=vec3(4, 1.2, -1), radius=1.8, material=S.dielectric(ri=1.5)), density=0.5, albedo=T.constant(color=vec3(0.7, 0.8, 0.9)))
    list2 = BoxedHitlist(items=[ball(4, -1) for _ in range(3)] + [BoxedHitlist(items=[ball(4, -1), mist])] + [ball(4, 0) for _ in range(2)])  # a nested Hitlist splices in
    haze = H.constant_medium(boundary=H.sphere(center=vec3(0, 0, 0), radius=60, material=S.dielectric(ri=1.5)), density=0.01, albedo=T.constant(color=vec3(1, 1, 1)))
    sky = H.sphere(center=vec3(0, 0, 0), radius=500, material=S.diffuse_light(tex=T.constant(color=vec3(0.8, 0.9, 1.0))))
    ground = H.sphere(center=vec3(0, -1000, 0), radius=1000, material=S.lambertian(albedo=T.checkerboard(tex0=T.constant(color=vec3(0.2, 0.3, 0.1)), tex1=T.constant(color=vec3(0.9, 0.9, 0.9)), scale=10)))
    loose = H.make_bvh([ball(0, 0) for _ in range(10)] + [sky, ground], 0.0, 1.0, r.util.SplitMix64(3))
    hz = H.make_bvh([haze], 0.0, 1.0, r.util.SplitMix64(4))  # bvh-node(haze, haze): asked twice
    right = H.bvh_node(list2, hz, union([list2, hz]))
    world = H.bvh_node(H.bvh_node(list1, loose, union([list1, loose])), right, union([list1, loose, right]))
    cam = r.camera.thin_lens_camera(lookfrom=vec3(2, 4, 12), lookat=vec3(0, 1, 0), vup=vec3(0, 1, 0), vfov=40, aspect=2.0, aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0)
    return {"camera": cam, "world": world}


def test_hitlists_holding_media_below_bvh_nodes_match_the_nested_oracle(oracle):
    """RTMI_MEDIA_NARROWED (rtmi_scene_set_media_calls_narrowed): every media call carries the first primitive of the Hitlist items that narrow its t-max; the
    bvh-nodes above pass the interval on un-narrowed.  Against the oracle's nested evaluation: segment logs, frames, BVH = flat scan; the narrowing must matter
    (the same calls declared un-narrowed give another frame); a clone carries the narrowing."""
    from oracle.tree import flatten_with_tree
    from tests.test_gpu_round3 import rms, RMS_TOL
    f = flatten_with_tree(narrowed_media_scene())
    calls, lo = [int(v) for v in f.media_calls], [int(v) for v in f.media_narrow_from]
    assert f.media_mode == 2 and len(calls) == 5 and calls[3] == calls[4] and lo[3] == calls[3] == lo[4]            # the haze: twice, un-narrowed
    assert lo[0] == lo[1] < calls[0] < calls[1] and lo[2] < calls[2] and len({lo[0], lo[2], lo[3]}) == 3            # fog + smoke share list 1, the mist has list 2
    nx, ny, ns = 64, 32, 8
    exp_lin, exp_q, exp_cnt = oracle.render(f, nx, ny, ns, 50, 0x5EED0002, nthreads=16)
    rng = np.random.default_rng(2)
    n = 4096
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    cam = oracle.probe_camera(f, rng.random((n, 2)), keys)
    ctr0 = int(cam[:, 7].max())
    ergb, enseg, elog, enlog = oracle.probe_paths(f, cam[:, :7], keys, depth=50, ctr0=ctr0, max_seg=6)
    logged = np.arange(6)[None, :] < enlog[:, None]
    for m in set(calls):
        assert (elog[:, :, 0][logged].astype(int) == m).sum() > 5, "paths must scatter in every medium (%d)" % m
    ctx = core.Context(0)
    ds = core.DeviceScene(f, ctx=ctx)
    frames = {}
    for accel in (1, 0):
        ctx.set_option("accel", accel)
        rgb, nseg, log, nlog = ds.probe_paths(cam[:, :7], keys, depth=50, ctr0=ctr0, max_seg=6)
        same = nseg == enseg
        assert same.mean() > 0.995, accel
        assert np.array_equal(log[same][:, :, 0], elog[same][:, :, 0]) and np.allclose(log[same], elog[same], rtol=1e-9, atol=1e-9), accel
        lin, q, cnt = ds.render(nx, ny, ns)
        assert abs(int(cnt[0]) - int(exp_cnt[0])) <= 52 and rms(lin, exp_lin) <= RMS_TOL, (accel, rms(lin, exp_lin), cnt, exp_cnt)
        frames[accel] = (lin, cnt)
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1]), "BVH and flat scan: the same frame"
    ctx2 = core.Context(0)
    twin = ds.clone(ctx2)  # rtmi_scene_clone replays the narrowed call sequence
    lin2, q2, cnt2 = twin.render(nx, ny, ns)
    assert np.array_equal(lin2, frames[0][0]) and np.array_equal(cnt2, frames[0][1])
    twin.close(); ctx2.close()
    c = np.ascontiguousarray(f.media_calls, np.int32)
    core.check(r._ffi.lib().rtmi_scene_set_media_calls(ds.handle, len(c), core.ptr(c)))  # the same calls, un-narrowed: a make-bvh world
    other, _, ocnt = ds.render(nx, ny, ns)
    assert int(ocnt[0]) != int(frames[1][1][0]), "the narrowing must matter in this scene"
    bad = np.ascontiguousarray(c + 1, np.int32)
    assert r._ffi.lib().rtmi_scene_set_media_calls_narrowed(ds.handle, len(c), core.ptr(c), core.ptr(bad)) != 0  # narrow_from beyond the medium
    ds.close(); ctx.close()


def test_tight_boxes_of_instanced_spheres_hold_on_silhouette_rays(monkeypatch):
    """Round 4 bounds a sphere under Translate / RotateY wrappers by the box of the sphere about its MAPPED centre (centre +- r) instead of the box of its local box's
    eight rotated corners.  The float traversal is only a filter in front of the exact test in the instance's frame, so the box must catch every ray that test
    calls a hit: rays aimed at the silhouettes of make-final's thousand instanced spheres -- tangent within a few ulps inside and outside, from near and far,
    along the axes and obliquely -- must give the tree the hits the flat scan gives, bit for bit."""
    import math
    flat = fl.flatten(r.scene.make_final(64, 64))
    inst = np.flatnonzero((flat.prim_kind[:flat.n_prims] == fl.PRIM_SPHERE) & (flat.prim_xform[:flat.n_prims, 1] > 0))
    assert len(inst) == 1000
    rng = np.random.default_rng(21)
    pick = rng.choice(inst, 6000)
    first, count = flat.prim_xform[pick[0]]
    assert count == 2 and np.all(flat.prim_xform[inst] == [first, count])
    # outward map of the chain (innermost wrapper first): RotateY (sin, cos) then Translate
    (k0, p0), (k1, p1) = [(int(flat.xform_kind[first + q]), flat.xform_param[first + q]) for q in range(2)]
    assert k0 == fl.XFORM_TRANSLATE and k1 == fl.XFORM_ROTATE_Y
    c = flat.prim_geom[pick, :3].copy()
    sn, cs = p1[0], p1[1]
    cw = np.stack([cs * c[:, 0] + sn * c[:, 2], c[:, 1], -sn * c[:, 0] + cs * c[:, 2]], axis=1) + p0[None, :]
    rad = flat.prim_geom[pick, 3]
    d = rng.normal(0, 1, (len(pick), 3))
    d[::5] = np.eye(3)[rng.integers(0, 3, len(d[::5]))] * rng.choice([-1.0, 1.0], len(d[::5]))[:, None] + rng.normal(0, 1e-9, (len(d[::5]), 3))  # nearly axis-parallel
    d /= np.linalg.norm(d, axis=1)[:, None]
    perp = np.cross(d, rng.normal(0, 1, d.shape))
    perp /= np.linalg.norm(perp, axis=1)[:, None]
    miss = rad * (1.0 + rng.choice([-1e-13, -1e-15, 0.0, 1e-15, 1e-13, 1e-9, -1e-9, 1e-6, -1e-6], len(pick)))   # tangent distance: inside / on / outside the silhouette
    dist = rng.choice([0.5, 3.0, 30.0, 3000.0], len(pick))  # (the cluster is dense: only rays that start close to the tangent point meet their own sphere first)
    o = cw + perp * miss[:, None] - d * dist[:, None]
    rays = np.concatenate([o, d * rng.uniform(0.5, 2.0, (len(pick), 1)), rng.uniform(0, 1, (len(pick), 1))], axis=1)
    monkeypatch.setenv("RTMI_FLAT_BELOW", "0")
    ctx = core.default_context()
    out = {}
    for accel in (1, 0):
        ctx.set_option("accel", accel)
        ds = core.DeviceScene(flat)
        out[accel] = ds.probe_hit(rays, 0.001, 3.4028234663852886e38)
        ds.close()
    ctx.set_option("accel", 1)
    assert np.array_equal(out[0], out[1], equal_nan=True)
    hit_target = (out[0][:, 0] > 0) & (out[0][:, 1] == pick)
    assert 0.05 < hit_target.mean() < 0.6 and (out[0][:, 0] > 0).mean() > 0.9  # the rays do graze: a good share end on the very sphere they were aimed at, not all


def test_mixed_kind_ties_fold_like_the_hitlist(oracle):
    """The branch-free fold of candidates into the closest-hit state (ext_update) against Hitlist.hit? (hitable.clj:15-26) where candidates TIE: a sphere
    replaces the closest hit only if t < closest (hitable.clj:195), a rectangle if t <= closest (hitable.clj:278) -- among primitives tied at the minimal t the
    last inclusive one after the first tied primitive wins, else the first.  The ray (0,0,0) -> (0,0,-1) meets the unit sphere about (0,0,-2) and the
    rectangle z = -1 at t = 1 exactly; every order of spheres and rectangles, through the small-world scan, the culled scan and the tree, and with
    t-max = 1 exactly (a rectangle at the caller's t-max is a hit, a sphere is not)."""
    from oracle.tree import flatten_with_tree
    import itertools
    H, S, T = r.hitable, r.shader, r.texture
    vec3 = lambda *a: np.array(a, np.float64)
    cam = r.camera.pinhole_camera(lookfrom=vec3(0, 0, 0), lookat=vec3(0, 0, -1), vup=vec3(0, 1, 0), vfov=40, aspect=1.0)
    def mat(k):
        return S.lambertian(albedo=T.constant(color=vec3(0.1 * (k + 1), 0.5, 0.5)))
    def sphere(k):
        return H.sphere(center=vec3(0, 0, -2), radius=1.0, material=mat(k))
    def rect(k):
        return H.rect_xy(x0=-1, y0=-1, x1=1, y1=1, k=-1.0, material=mat(k))
    far = [H.sphere(center=vec3(40 + 3 * k, 0, -50), radius=1.0, material=mat(9)) for k in range(3)]  # bystanders: the worlds are not all ties
    rays = np.array([[0, 0, 0, 0, 0, -1, 0.0], [0, 0, 0, 0, 0, -2, 0.0], [0.25, -0.5, 0, 0, 0, -1, 0.0], [0, 0, 0, 0.1, 0, -1, 0.0]])
    n_checked = 0
    for pattern in itertools.product("sr", repeat=4):
        items = [sphere(k) if c == "s" else rect(k) for k, c in enumerate(pattern)] + far
        for world in (H.hitlist(items=items), H.make_bvh(items, 0.0, 1.0)):
            f = flatten_with_tree({"camera": cam, "world": world})
            for tmax in (3.4028234663852886e38, 1.0, 0.5):
                exp = oracle.probe_hit(f, rays, tmax=tmax)
                ctx = core.Context(0)
                ds = core.DeviceScene(f, ctx=ctx)
                got = {}
                for tag, opts in (("small", {"accel": 0}), ("tree", {"accel": 1, "flat_below": 0})):
                    for k, v in opts.items():
                        ctx.set_option(k, v)
                    got[tag] = ds.probe_hit(rays, t_max=tmax)
                ds.close(); ctx.close()
                os.environ["RTMI_SMALL_SCAN"] = "0"
                try:
                    ctx = core.Context(0)
                    ds = core.DeviceScene(f, ctx=ctx)
                    ctx.set_option("accel", 0)
                    got["culled"] = ds.probe_hit(rays, t_max=tmax)
                    ds.close(); ctx.close()
                finally:
                    del os.environ["RTMI_SMALL_SCAN"]
                assert np.array_equal(got["small"], got["tree"]) and np.array_equal(got["small"], got["culled"]), (pattern, tmax)
                if isinstance(world, H.Hitlist):  # (among bvh-node siblings the reference lets the RIGHT child win a tie, hitable.clj:103-105; the device folds a
                    assert np.array_equal(got["small"][:, :9], exp[:, :9]), (pattern, tmax)  # flattened world by the Hitlist rule: exact ties between leaves are the one place they differ)
                # the rule itself, for the ray along the axis (every candidate ties at t = 1): the last rectangle if there is one, else the first sphere
                if isinstance(world, H.Hitlist):
                    first = got["small"][0]
                    last_rect = max([k for k, c in enumerate(pattern) if c == "r"], default=-1)
                    if tmax >= 1.0 and (last_rect >= 0 or tmax > 1.0):
                        assert first[0] == 1.0 and first[2] == 1.0 and int(first[1]) == (last_rect if last_rect >= 0 else 0), (pattern, tmax, first[:3])
                    else:
                        assert first[0] == 0.0, (pattern, tmax, first[:3])
                n_checked += 1
    assert n_checked == 16 * 2 * 3


def test_turbulence_served_by_the_wave_is_bit_identical(oracle):
    """perlin_turbulence_wave (a Marble / PerlinTurbulence texture's turbulence evaluated by the whole wave when at most four of its lanes ask: one (octave, corner)
    term per lane through LDS, added up in perlin.clj's order) against the lanes' own loops (what a wave runs when many lanes ask) and, for PerlinTurbulence (no
    sin), the oracle: Shader.scatter's attenuation of the same hit records, bit for bit, whether they come one to four per launch (the wave serves them; with
    three active lanes every lane evaluates eleven terms) or 64 per wave.  Depths 1, 4, 7 (two rounds of four octaves) and 9 (three)."""
    T, S, H = r.texture, r.shader, r.hitable
    vec3 = lambda *a: np.array(a, np.float64)
    texs = [T.perlin_turbulence(scale=4, depth=7), T.perlin_turbulence(scale=0.7, depth=1), T.perlin_turbulence(scale=2.5, depth=9), T.perlin_turbulence(scale=1.0, depth=4),
            T.marble(scale=0.1, depth=4), T.marble(scale=3.0, depth=7)]
    world = H.hitlist(items=[H.sphere(center=vec3(3 * k, 0, 0), radius=1.0, material=S.lambertian(albedo=t)) for k, t in enumerate(texs)])
    f = fl.flatten(world, r.camera.PinholeCamera(*(np.zeros(3),) * 4))
    rng = np.random.default_rng(17)
    n = 256
    p = rng.normal(0, 40, (n, 3))
    p[:16] = np.round(p[:16])  # lattice points
    nrm = rng.normal(0, 1, (n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    hits = np.concatenate([p, nrm, rng.random((n, 2))], axis=1)
    rays = np.concatenate([p - nrm, -nrm + 0.1 * rng.normal(0, 1, (n, 3)), rng.random((n, 1))], axis=1)
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    ds = core.DeviceScene(f)
    for m, t in enumerate(texs):
        dense = ds.probe_scatter(m, rays, hits, keys)  # 64 requests per wave: every lane's own loop
        assert np.all(dense[:, 0] == 1.0)
        for lo, cnt in ((0, 1), (1, 2), (3, 3), (6, 4), (100, 3), (250, 1)):  # <= 4 requests in the launch's one wave: served by the wave
            few = ds.probe_scatter(m, rays[lo:lo + cnt], hits[lo:lo + cnt], keys[lo:lo + cnt])
            assert np.array_equal(few, dense[lo:lo + cnt]), (type(t).__name__, t.depth, lo, cnt)
        five = ds.probe_scatter(m, rays[10:15], hits[10:15], keys[10:15])  # five requests: the lanes' own loops again
        assert np.array_equal(five, dense[10:15])
        if isinstance(t, T.PerlinTurbulence):
            exp = oracle.probe_scatter(f, m, rays, hits, keys)
            assert np.array_equal(dense, exp), (t.scale, t.depth)
    ds.close()


def test_mixed_kind_soak_short():
    """scripts/gpu_soak_ext.py at a size that takes seconds: random mixed-kind worlds, adversarial rays (corners, edges, plane origins, axis-parallel, huge / tiny /
    non-finite), every device path the same bits, Hitlist worlds = the nested oracle, bvh worlds = the Hitlist semantics"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_soak_ext.py"), "30000", "6"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "SOAK-EXT PASS" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]

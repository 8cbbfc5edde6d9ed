"""GPU tests added in round 3: the in-library RCCL gather executed on the hardware that exists (a one-rank communicator), which gather
ran, the error path of the multi-device entry, back-to-back asynchronous frames, the sample-buffer budget under memory pressure,
BASELINE's C4 at full size, a many-pass C5, the traversal counters of the mixed-kind (f3 / f4) kernels.  Everything goes through the C-ABI."""
import ctypes as C
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
GOLD = os.path.join(ROOT, "tests", "golden")
RMS_TOL = 1e-4


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def gather_path(ctx):
    p = C.c_int32(-1)
    core.check(r._ffi.lib().rtmi_last_gather_path(ctx.handle, C.byref(p)))
    return r._ffi.GATHER_PATHS[p.value]


def stream_idle(ctx):
    v = C.c_int32(-1)
    core.check(r._ffi.lib().rtmi_stream_idle(ctx.handle, C.byref(v)))
    return bool(v.value)


def last_passes(ctx):
    v = C.c_int32(-1)
    core.check(r._ffi.lib().rtmi_last_passes(ctx.handle, C.byref(v)))
    return v.value


# ---- the RCCL branch of rtmi_render_multi_device, on one GPU ------------------------------------------------------------------------
def test_rccl_gather_one_rank_communicator(cover11, monkeypatch):
    """RTMI_MULTI_GATHER=rccl with ONE replica runs the whole in-library RCCL path -- dlopen of librccl (in this process: the copy PyTorch
    already maps), the six entry points, ncclCommInitAll over one device, ncclGroupStart / in-place ncclGather / ncclGroupEnd on the
    context stream, assemble -- which a one-GPU host otherwise never reaches.  Image and counters equal rtmi_render's; the context
    reports the path that ran.  (A communicator over several devices needs several devices: not on this pool.)"""
    from raytrace_clj_amd import dist as rdist
    nx, ny, ns = 200, 100, 8
    flat = fl.flatten(cover11)
    ds = core.DeviceScene(flat)
    lin, q, cnt = ds.render(nx, ny, ns)
    ds.close()
    monkeypatch.setenv("RTMI_MULTI_GATHER", "rccl")
    one = rdist.MultiDevice(flat, [0])
    for _ in range(3):  # the communicator is created once and kept
        mlin, mq, mcnt = one.render(nx, ny, ns)
        assert np.array_equal(mlin, lin) and np.array_equal(mq, q) and np.array_equal(mcnt, cnt)
        assert gather_path(one.ctxs[0]) == "rccl"
    assert one.last_gather_ms() >= 0.0
    one.close()
    # RCCL refuses one device twice in a communicator: forcing it onto replicas that share a device is an argument error, not a substitution
    two = rdist.MultiDevice(flat, [0, 0])
    with pytest.raises(core.RtmiError) as e:
        two.render(nx, ny, ns)
    assert e.value.code == -1 and "distinct devices" in str(e.value)
    monkeypatch.delenv("RTMI_MULTI_GATHER")
    mlin, mq, mcnt = two.render(nx, ny, ns)  # the same replicas, default policy: device copies
    assert np.array_equal(mlin, lin) and np.array_equal(mcnt, cnt) and gather_path(two.ctxs[0]) == "same-device"
    two.close()
    plain = rdist.MultiDevice(flat, [0])
    plain.render(nx, ny, ns)
    assert gather_path(plain.ctxs[0]) == "none"
    plain.close()


def test_rccl_unusable_is_an_error_when_forced_not_a_substitution(tmp_path):
    """RTMI_MULTI_GATHER=rccl with an RCCL that cannot be opened ($RTMI_RCCL_LIB names a missing library): the call fails with the loader's
    message (the path that used to dereference dlerror()'s second, NULL, result) -- in a process of its own, because the library gives up
    on RCCL for the rest of the process once it could not be opened."""
    code = (
        "import numpy as np, raytrace_clj_amd as r\n"
        "from raytrace_clj_amd import core, flatten as fl, dist as rdist\n"
        "flat = fl.flatten(r.scene.make_random_scene(200, 100, 3, False))\n"
        "md = rdist.MultiDevice(flat, [0])\n"
        "try:\n"
        "    md.render(64, 32, 2)\n"
        "    print('NO ERROR')\n"
        "except core.RtmiError as e:\n"
        "    print('ERR', e.code, str(e))\n"
    )
    env = dict(os.environ, RTMI_MULTI_GATHER="rccl", RTMI_RCCL_LIB="librccl_does_not_exist.so.1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ERR -2" in out.stdout and "dlopen(librccl_does_not_exist.so.1)" in out.stdout, out.stdout + out.stderr


def test_torch_free_c_host_runs_the_rccl_gather(tmp_path):
    """tests/host_smoke.c with RTMI_MULTI_GATHER=rccl: a process that has never seen torch opens /opt/rocm's librccl.so.1 by soname, binds
    the entry points and runs the one-rank gather (third frame of its output) -- what a JVM host would load."""
    z = np.load(os.path.join(GOLD, "render_cover_n3.npz"))
    exe = tmp_path / "host_smoke"
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", str(exe), os.path.join(ROOT, "tests", "host_smoke.c"), "-ldl"], check=True)
    nx, ny = int(z["nx"]), int(z["ny"])
    with open(tmp_path / "scene.bin", "wb") as f:
        f.write(np.array([len(z["prim_kind"]), len(z["mat_kind"]), len(z["tex_kind"]), int(z["cam_kind"]), nx, ny, int(z["ns"]), int(z["depth"])], np.int32).tobytes())
        f.write(np.array([int(z["seed"])], np.uint64).tobytes())
        for k, dt in (("prim_kind", np.int32), ("prim_geom", np.float64), ("prim_mat", np.int32), ("mat_kind", np.int32), ("mat_tex", np.int32),
                      ("mat_param", np.float64), ("tex_kind", np.int32), ("tex_param", np.float64), ("tex_child", np.int32), ("cam", np.float64)):
            f.write(np.ascontiguousarray(z[k], dt).tobytes())
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "LD_PRELOAD")}
    env["RTMI_HOST_SMOKE_RCCL"] = "1"
    out = subprocess.run([str(exe), r._ffi.LIB_PATH, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rccl probe ok" in out.stdout and "gather path of the one-rank frame: 3" in out.stdout, out.stdout + out.stderr
    raw = open(tmp_path / "out.bin", "rb").read()
    n = nx * ny * 3
    one = n * 8 + n + 16
    assert len(raw) == 3 * one
    blob = raw[2 * one:]
    lin = np.frombuffer(blob[:n * 8], np.float64).reshape(ny, nx, 3)
    cnt = np.frombuffer(blob[n * 8 + n:], np.uint64)
    assert rms(lin, z["linear"]) < 1e-13 and np.array_equal(cnt, z["counters"])


# ---- error path: a replica fails after earlier replicas were launched -------------------------------------------------------------------
def test_multi_device_error_leaves_no_work_in_flight(cover11):
    """replica 2's render fails (test hook) after replicas 0 and 1 have been enqueued: the call returns the error only once every stream
    it touched is idle (the caller may destroy the contexts next), and the replicas render correctly afterwards"""
    import torch
    from raytrace_clj_amd import dist as rdist
    nx, ny, ns = 800, 400, 64  # ~1 ms per replica: long enough to still be running when the failing replica returns
    flat = fl.flatten(r.scene.make_random_scene(nx, ny, 11, False))
    ds = core.DeviceScene(flat)
    lin, q, cnt = ds.render(nx, ny, ns)
    ds.close()
    md = rdist.MultiDevice(flat, [0, 0, 0])
    dl = torch.zeros((ny, nx, 3), dtype=torch.float64, device="cuda")
    dc = torch.zeros(2, dtype=torch.int64, device="cuda")
    md.render_device(nx, ny, ns, dl, None, dc)  # workspaces allocated
    md.sync()
    for victim in (2, 1):
        md.ctxs[victim].set_option("test_fail_next_render", 1)
        with pytest.raises(core.RtmiError) as e:
            md.render_device(nx, ny, ns, dl, None, dc)
        assert e.value.code == -2 and "test hook" in str(e.value)
        assert all(stream_idle(c) for c in md.ctxs), "streams must be idle when the error is returned"
    md.render_device(nx, ny, ns, dl, None, dc)
    md.sync()
    assert np.array_equal(dl.cpu().numpy(), lin) and [int(dc[0]), int(dc[1])] == [int(cnt[0]), int(cnt[1])]
    md.close()


def test_back_to_back_async_multi_frames_do_not_overwrite_each_other():
    """rtmi_render_multi_device is asynchronous: frame k+1's render into a replica's record must wait for frame k's copy out of it (the copy
    runs on replica 0's stream).  Six frames with six seeds, enqueued without a host synchronisation, each into its own output."""
    import torch
    from raytrace_clj_amd import dist as rdist
    nx, ny, ns = 400, 200, 16
    flat = fl.flatten(r.scene.make_random_scene(nx, ny, 11, False))
    seeds = [0x5EED0002 + 977 * k for k in range(6)]
    ds = core.DeviceScene(flat)
    want = [ds.render(nx, ny, ns, seed=s)[0] for s in seeds]
    ds.close()
    md = rdist.MultiDevice(flat, [0, 0, 0, 0])
    outs = [torch.zeros((ny, nx, 3), dtype=torch.float64, device="cuda") for _ in seeds]
    torch.cuda.synchronize()
    for s, o in zip(seeds, outs):
        md.render_device(nx, ny, ns, o, None, None, seed=s)
    md.sync()
    for k, (o, w) in enumerate(zip(outs, want)):
        assert np.array_equal(o.cpu().numpy(), w), "frame %d" % k
    md.close()


# ---- the sample-buffer budget ---------------------------------------------------------------------------------------------------------
def test_sample_buffer_allocation_failure_means_more_passes_not_an_error(cover11):
    """the per-sample colour buffer: when its allocation fails the render halves the pass and retries (test hook: the next two allocations
    fail), and the image does not depend on the split"""
    nx, ny, ns = 320, 160, 64
    flat = fl.flatten(cover11)
    ctx = core.Context(0, timing=True)
    ds = core.DeviceScene(flat, ctx=ctx)
    base = ds.render(nx, ny, ns)
    assert last_passes(ctx) == 1
    ds.close(); ctx.close()
    ctx = core.Context(0, timing=True)
    ds = core.DeviceScene(flat, ctx=ctx)
    ctx.set_option("test_fail_allocs", 2)
    again = ds.render(nx, ny, ns)
    assert last_passes(ctx) == 4 and ctx.last_trace_ms()[1] == 4
    for x, y in zip(base, again):
        assert np.array_equal(x, y)
    ctx.set_option("test_fail_allocs", 64)  # nothing can be allocated at all: the error surfaces, as RTMI_E_NOMEM
    ds2 = core.DeviceScene(flat, ctx=ctx)
    ctx.set_option("workspace_bytes", 1 << 40)
    with pytest.raises(core.RtmiError) as e:
        ds2.render(nx * 2, ny * 2, ns)  # a larger frame: the buffer has to grow
    assert e.value.code == -4
    ctx.set_option("test_fail_allocs", 0)
    ok = ds2.render(nx, ny, ns)
    assert np.array_equal(ok[0], base[0])
    ds2.close(); ds.close(); ctx.close()


def test_sample_buffer_budget_follows_free_hbm():
    """a host that holds most of the HBM itself (here: one torch tensor) still gets its frame: the budget is clamped to what is free, the
    frame takes more passes, the image is the same"""
    import torch
    nx, ny, ns = 1920, 1080, 64  # 49.8 MB per sample: 3.2 GB in one pass
    flat = fl.flatten(r.scene.make_random_scene(nx, ny, 11, False))
    ctx = core.Context(0)
    ds = core.DeviceScene(flat, ctx=ctx)
    base = ds.render(nx, ny, ns)
    assert last_passes(ctx) == 1
    ds.close(); ctx.close()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    leave = 2 << 30
    hog = torch.empty(int(free - leave), dtype=torch.uint8, device="cuda")
    try:
        ctx = core.Context(0)
        ds = core.DeviceScene(flat, ctx=ctx)
        lin, q, cnt = ds.render(nx, ny, ns)
        passes = last_passes(ctx)
        ds.close(); ctx.close()
    finally:
        del hog
        torch.cuda.empty_cache()
    assert passes >= 2, passes
    assert np.array_equal(lin, base[0]) and np.array_equal(q, base[1]) and np.array_equal(cnt, base[2])


# ---- BASELINE configs[3] at full size ---------------------------------------------------------------------------------------------------
def test_config_c4_full_size(oracle):
    """BASELINE configs[3]: 3840x2160x512spp, cover scene n=11, on ONE GPU at the full sample count: determinism, sample-pass split
    invariance (2 passes by default, 7 through a 16 GiB budget), an 8x4 oracle region at 512 spp that is also the crop of the frame, and the
    frame dealt over 8 replicas (the 8-GPU partition, replicas sharing this GPU: device-copy gather) with counters summed.  What stays
    untested here is only the xGMI gather between distinct devices."""
    from raytrace_clj_amd import dist as rdist
    nx, ny, ns = 3840, 2160, 512
    flat = fl.flatten(r.scene.make_random_scene(nx, ny, 11, False))
    ctx = core.Context(0)
    ds = core.DeviceScene(flat, ctx=ctx)
    base, q, cnt = ds.render(nx, ny, ns)
    p_default = last_passes(ctx)
    ctx.set_option("workspace_bytes", 16 << 30)
    split, q2, cnt2 = ds.render(nx, ny, ns)
    p_split = last_passes(ctx)
    assert p_default < p_split and p_split >= 6, (p_default, p_split)
    assert np.array_equal(base, split) and np.array_equal(q, q2) and np.array_equal(cnt, cnt2), "deterministic, whatever the pass split"
    assert cnt[1] == nx * ny and 1.5 * nx * ny * ns < cnt[0] < 6 * nx * ny * ns
    region = (1900, 1400, 1908, 1404)
    lin, qr, cr = ds.render(nx, ny, ns, region=region)
    exp, eq, ecnt = oracle.render(flat, nx, ny, ns, 50, core.RENDER_SEED, region=region, nthreads=64)
    assert rms(lin, exp) <= RMS_TOL and rms(lin, exp) < 1e-13 and np.array_equal(cr, ecnt)
    assert np.array_equal(lin, base[1400:1404, 1900:1908])
    ds.close(); ctx.close()
    md = rdist.MultiDevice(flat, [0] * 8)
    mlin, mq, mcnt = md.render(nx, ny, ns)
    md.close()
    assert np.array_equal(mlin, base) and np.array_equal(mq, q) and np.array_equal(mcnt, cnt)
    assert np.isfinite(base).all() and base.min() >= 0


def test_config_c5_many_passes(oracle):
    """BASELINE configs[4] (1920x1080x4096spp, 80 % glass) through an 8 GiB sample buffer: 24+ sample passes with running sums carried
    between them, equal to the default split (3 passes); one oracle region at the full 4096 spp."""
    nx, ny, ns = 1920, 1080, 4096
    flat = fl.flatten(r.scene.make_random_scene(nx, ny, 11, False, mix=(0.1, 0.2)))
    ctx = core.Context(0)
    ds = core.DeviceScene(flat, ctx=ctx)
    base, q, cnt = ds.render(nx, ny, ns)
    p_default = last_passes(ctx)
    ctx.set_option("workspace_bytes", 8 << 30)
    many, q2, cnt2 = ds.render(nx, ny, ns)
    p_many = last_passes(ctx)
    assert p_default <= 4 and p_many >= 24, (p_default, p_many)
    assert np.array_equal(base, many) and np.array_equal(q, q2) and np.array_equal(cnt, cnt2)
    assert cnt[1] == nx * ny and 2.5 * nx * ny * ns < cnt[0] < 8 * nx * ny * ns
    region = (1000, 700, 1008, 704)
    lin, qr, cr = ds.render(nx, ny, ns, region=region)
    exp, eq, ecnt = oracle.render(flat, nx, ny, ns, 50, core.RENDER_SEED, region=region, nthreads=64)
    assert rms(lin, exp) < 1e-13 and np.array_equal(cr, ecnt) and np.array_equal(lin, base[700:704, 1000:1008])
    ds.close(); ctx.close()


# ---- metrics.clj:10 aabb.intersection.total for the mixed-kind (f3 / f4) kernels ------------------------------------------------------------
def test_traversal_counters_of_the_mixed_kind_kernels():
    """rtmi_last_traversal_counters for scenes the EXT instantiations render (rectangles, instances, media, procedural textures): the
    counting instantiation renders the same image, the counts are deterministic, independent of the time-slicing threshold and of the sample
    pass split, and of a plausible size (a few to a few dozen node visits per segment)"""
    for name, scene, nx, ny, ns in (("cornell", r.scene.make_cornell_box(64, 64), 64, 64, 8), ("final", r.scene.make_final(96, 96), 96, 96, 4)):
        flat = fl.flatten(scene)
        ref = None
        for lanes in (8, 0, 33):
            ctx = core.Context(0)
            ctx.set_option("suspend_lanes", lanes)
            ds = core.DeviceScene(flat, ctx=ctx)
            base = ds.render(nx, ny, ns)
            ctx.set_option("count_traversal", 1)
            counted = ds.render(nx, ny, ns)
            trav = ctx.last_traversal_counters()
            ctx.set_option("workspace_bytes", 1 << 20)
            ds.render(nx, ny, ns)
            assert ctx.last_traversal_counters() == trav, (name, "pass split")
            ds.close(); ctx.close()
            for x, y in zip(base, counted):
                assert np.array_equal(x, y), name
            rays = int(base[2][0])
            assert trav[0] % 2 == 0 and 2 * rays <= trav[0] < 400 * rays and rays <= trav[1] < 100 * rays, (name, trav, rays)
            if ref is None:
                ref = (base, trav)
            else:
                assert np.array_equal(base[0], ref[0][0]) and trav == ref[1], (name, lanes)


# ---- the entry grid of the BVH traversal ----------------------------------------------------------------------------------------------------------
def test_entry_grid_is_bit_identical_to_the_flat_scan(monkeypatch):
    """rtmi_device.h: bvh_grid_entry -- layer-like sphere scenes get a grid of per-cell BVHs and a ray whose clipped segment stays within a few
    cells starts its traversal there.  Whatever the grid's shape (RTMI_GRID = cells per side, RTMI_GRID_KMAX = cells a ray may touch), hits
    must equal the exact Hitlist scan's bit for bit -- on grazing rays, rays along cell borders, rays leaving sphere surfaces, silhouette rays --
    and so must whole renders and their counters, in both precisions and with moving / tall / overlapping spheres; and the grid must actually
    shorten traversals (node visits per segment), or it is not being used.  Segments that touch more than 2 x 2 cells are walked in pieces (the probes
    walk them in a loop, the render kernel parks the lane between pieces): the same grids with RTMI_GRID_WALK=0 send them to the root of the whole tree
    instead -- same bits, more node visits."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import grazing_rays, tangent_rays, random_rays, layer_scene
    def sweeps():
        """spheres that each sweep across the whole layer during the shutter: every one spans most cells, per-cell trees would multiply them --
        the library must decline the grid (and still be right)"""
        from raytrace_clj_amd.util import vec3
        rng = np.random.default_rng(8)
        H, S, T = r.hitable, r.shader, r.texture
        mat = S.lambertian(albedo=T.constant(color=vec3(0.5, 0.5, 0.5)))
        items = [H.sphere(center=vec3(0, -1000, 0), radius=1000, material=mat)]
        for k in range(400):
            c0 = vec3(rng.uniform(-20, 20), 0.2, rng.uniform(-20, 20))
            items.append(H.moving_sphere(center0=c0, t0=0.0, center1=c0 + vec3(rng.uniform(-40, 40), 0.0, rng.uniform(-40, 40)), t1=1.0, radius=0.2, material=mat))
        cam = r.camera.thin_lens_camera(lookfrom=vec3(20, 3, 10), lookat=vec3(0, 0, 0), vup=vec3(0, 1, 0), vfov=40, aspect=2.0, aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0)
        return {"camera": cam, "world": H.hitlist(items=items)}
    scenes = [("cover11m", r.scene.make_random_scene(160, 80, 11, True), True), ("cover30", r.scene.make_random_scene(160, 80, 30, False), True),
              ("layer", layer_scene(3, n=1200), True), ("sweeps", sweeps(), False)]
    for name, sc, expect_grid in scenes:
        flat = fl.flatten(sc)
        rays = np.concatenate([grazing_rays(flat, 60000, 11), tangent_rays(flat, 30000, 12), random_rays(10000, 13, spread=40.0)])
        ref = {}
        ctx = core.Context(0)
        ctx.set_option("accel", 0)
        ds = core.DeviceScene(flat, ctx=ctx)
        for prec in ("f64", "f32"):
            ref[prec] = (ds.probe_hit(rays, precision=prec), ds.probe_hit(rays, 0.0, 3.4028234663852886e38, precision=prec), ds.render(160, 80, 8, precision=prec))
        ds.close(); ctx.close()
        assert (ref["f64"][0][:, 0] == 1).mean() > 0.2, "the probe rays must hit things"
        visits = {}
        for spec in ("0:4", "1:4", "1:4:nowalk", "5:1", "23:2", "64:4", "40:4", "40:4:nowalk"):
            g, k = spec.split(":")[:2]
            monkeypatch.setenv("RTMI_GRID", g)
            monkeypatch.setenv("RTMI_GRID_KMAX", k)
            monkeypatch.setenv("RTMI_GRID_WALK", "0" if spec.endswith("nowalk") else "1")  # long segments: walked in pieces of 2 x 2 cells / from the root of the whole tree
            ctx = core.Context(0)
            ds = core.DeviceScene(flat, ctx=ctx)  # the grid is built with the scene
            for prec in ("f64", "f32"):
                a = ds.probe_hit(rays, precision=prec)
                b = ds.probe_hit(rays, 0.0, 3.4028234663852886e38, precision=prec)
                assert np.array_equal(a, ref[prec][0], equal_nan=True) and np.array_equal(b, ref[prec][1], equal_nan=True), (name, spec, prec)
                for lanes in (12, 0):
                    ctx.set_option("suspend_lanes", lanes)
                    out = ds.render(160, 80, 8, precision=prec)
                    for x, y in zip(out, ref[prec][2]):
                        assert np.array_equal(x, y), (name, spec, prec, lanes)
            ctx.set_option("count_traversal", 1)
            out = ds.render(160, 80, 8)
            visits[spec] = ctx.last_traversal_counters()[0] / 2 / float(out[2][0])
            ds.close(); ctx.close()
        if expect_grid:
            assert visits["1:4"] < 0.8 * visits["0:4"], (name, visits)
            assert visits["1:4"] < 0.97 * visits["1:4:nowalk"] and visits["40:4"] <= visits["40:4:nowalk"], (name, visits)  # the walk is taken, and it is shorter (40 cells per side: declined for the small scenes)
        else:
            assert visits["1:4"] == visits["0:4"] == visits["64:4"], (name, visits)  # declined: the whole tree, whatever the requested shape


# ---- ConstantMedium inside a Hitlist (hitable.clj:15-26 + 516-541) -------------------------------------------------------------------------------------
def hitlist_media_scene(seed=5, fog_twice=False):
    """a world that IS a Hitlist (no make-bvh): surfaces, a fog ball in the middle of the list, more surfaces -- some of them inside and behind
    the fog --, a second medium bounded by a Box, nested Hitlists and a translated, rotated box"""
    from raytrace_clj_amd.util import vec3
    rng = np.random.default_rng(seed)
    H, S, T = r.hitable, r.shader, r.texture
    grey = S.lambertian(albedo=T.constant(color=vec3(0.6, 0.6, 0.6)))
    items = [H.sphere(center=vec3(0, 0, 0), radius=500, material=S.diffuse_light(tex=T.constant(color=vec3(0.8, 0.9, 1.0)))),
             H.sphere(center=vec3(0, -1000, 0), radius=1000, material=S.lambertian(albedo=T.checkerboard(tex0=T.constant(color=vec3(0.2, 0.3, 0.1)), tex1=T.constant(color=vec3(0.9, 0.9, 0.9)), scale=10)))]
    def ball():
        return H.sphere(center=vec3(rng.uniform(-5, 5), rng.uniform(0.2, 2.5), rng.uniform(-5, 5)), radius=float(rng.uniform(0.2, 0.7)),
                        material=[grey, S.metal(albedo=T.constant(color=vec3(0.8, 0.7, 0.6)), fuzz=0.1), S.dielectric(ri=1.5)][int(rng.integers(0, 3))])
    items += [ball() for _ in range(12)]
    fog = H.constant_medium(boundary=H.sphere(center=vec3(0, 1.5, 0), radius=2.5, material=S.dielectric(ri=1.5)), density=0.35, albedo=T.constant(color=vec3(0.9, 0.9, 0.9)))
    items.append(fog)                                                         # narrowed by everything above, narrows everything below
    items.append(H.hitlist(items=[ball() for _ in range(6)]))                 # a nested Hitlist splices in
    items.append(H.translate(item=H.rotate_y(item=H.box(p0=vec3(0, 0, 0), p1=vec3(1, 2, 1), material=grey), theta=25.0), offset=vec3(1.5, 0, -2.0)))
    smoke = H.constant_medium(boundary=H.box(p0=vec3(-4, 0, 1), p1=vec3(-1, 2, 4), material=grey), density=0.8, albedo=T.constant(color=vec3(0.1, 0.1, 0.1)))
    items.append(smoke)
    items += [ball() for _ in range(8)]
    if fog_twice:
        items.append(fog)                                                     # the SAME record listed again: asked a second time, narrowed by everything above
        items += [ball() for _ in range(3)]
    cam = r.camera.thin_lens_camera(lookfrom=vec3(9, 3, 7), lookat=vec3(0, 1, 0), vup=vec3(0, 1, 0), vfov=40, aspect=2.0, aperture=0.0, focus_dist=10.0, t0=0.0, t1=1.0)
    return {"camera": cam, "world": H.hitlist(items=items)}


def test_media_inside_a_hitlist_match_the_nested_oracle(oracle):
    """A world built as a Hitlist with ConstantMedium items: Hitlist.hit? hands every item the t-max narrowed by the items before it, and a
    medium draws its random number inside hit?, so WHERE in the list it stands changes what it draws (round 2 rejected such worlds).  The
    device scans the list in pieces around its media (RTMI_MEDIA_HITLIST); the oracle evaluates the nested records the way the reference does
    (oracle/tree.py, node_hit).  Geometry logs equal (media hits within the log tolerance: ocml vs glibc log), images within the tolerance,
    BVH and flat scan alike -- and the narrowing must matter: declared as a bvh-descent world the same primitives give a different frame."""
    from oracle.tree import flatten_with_tree
    sc = hitlist_media_scene()
    f = flatten_with_tree(sc)
    assert f.media_mode == 1 and len(f.media_calls) == 2 and list(f.media_calls) == sorted(f.media_calls)
    nx, ny, ns = 64, 32, 8
    exp_lin, exp_q, exp_cnt = oracle.render(f, nx, ny, ns, 50, 0x5EED0002, nthreads=16)
    rng = np.random.default_rng(2)
    n = 4096
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    cam = oracle.probe_camera(f, rng.random((n, 2)), keys)
    ctr0 = int(cam[:, 7].max())
    ergb, enseg, elog, enlog = oracle.probe_paths(f, cam[:, :7], keys, depth=50, ctr0=ctr0, max_seg=6)
    media = set(int(m) for m in f.media_calls)
    logged = np.arange(6)[None, :] < enlog[:, None]
    assert np.isin(elog[:, :, 0][logged].astype(int), list(media)).sum() > 50, "the probe paths must scatter inside the media"
    ctx = core.Context(0)
    ds = core.DeviceScene(f, ctx=ctx)
    frames = {}
    for accel in (1, 0):
        ctx.set_option("accel", accel)
        rgb, nseg, log, nlog = ds.probe_paths(cam[:, :7], keys, depth=50, ctr0=ctr0, max_seg=6)
        same = nseg == enseg
        assert same.mean() > 0.995, accel
        assert np.array_equal(log[same][:, :, 0], elog[same][:, :, 0]) and np.allclose(log[same], elog[same], rtol=1e-9, atol=1e-9), accel
        assert np.allclose(rgb[same], ergb[same], atol=1e-9, rtol=0)
        lin, q, cnt = ds.render(nx, ny, ns)
        assert abs(int(cnt[0]) - int(exp_cnt[0])) <= 52 + 1e-5 * int(exp_cnt[0]) and rms(lin, exp_lin) <= RMS_TOL, (accel, rms(lin, exp_lin), cnt, exp_cnt)
        frames[accel] = (lin, cnt)
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1]), "BVH and flat scan: the same frame"
    core.check(r._ffi.lib().rtmi_scene_set_media_mode(ds.handle, 0))  # the same primitives read as a make-bvh world: un-narrowed media
    other, _, ocnt = ds.render(nx, ny, ns)
    assert not np.array_equal(other, frames[1][0]) and int(ocnt[0]) != int(frames[1][1][0]), "the narrowing must matter in this scene"
    with pytest.raises(core.RtmiError):
        core.check(r._ffi.lib().rtmi_scene_set_media_mode(ds.handle, 7))
    ds.close(); ctx.close()


# ---- small mixed-kind scenes: the scan answers for the tree ----------------------------------------------------------------------------
def test_small_mixed_scenes_are_scanned_even_when_the_tree_is_asked_for(monkeypatch):
    """Option "flat_below" (default 24): a Cornell box's 18 primitives render 8 % faster through the flat scan than through a tree, so the
    library scans them even under accel = BVH -- same image, same counters, and rtmi_last_accel says so.  The tree still runs when its
    traversal counters are asked for, for larger mixed-kind scenes and for sphere-only scenes of any size."""
    monkeypatch.delenv("RTMI_FLAT_BELOW", raising=False)  # (conftest switches the shortcut off for every other test)
    nx, ny, ns = 96, 96, 6
    cb = fl.flatten(r.scene.make_cornell_box(nx, ny))
    assert cb.n_prims < 24
    out = {}
    for below in (None, 0, 1 << 20):
        ctx = core.Context(0)
        ctx.set_option("accel", 1)
        if below is not None:
            ctx.set_option("flat_below", below)
        ds = core.DeviceScene(cb, ctx=ctx)
        out[below] = ds.render(nx, ny, ns) + (ctx.last_accel(),)
        if below is None:  # the counting instantiation walks the tree
            ctx.set_option("count_traversal", 1)
            cnt = ds.render(nx, ny, ns)
            assert ctx.last_accel() == "bvh" and ctx.last_traversal_counters()[0] > 0
            assert np.array_equal(cnt[0], out[None][0])
        ds.close(); ctx.close()
    assert out[None][3] == "flat" and out[0][3] == "bvh" and out[1 << 20][3] == "flat"
    for k in (0, 1 << 20):
        assert np.array_equal(out[None][0], out[k][0]) and np.array_equal(out[None][1], out[k][1]) and list(out[None][2]) == list(out[k][2])
    # the flat scan when it was asked for
    ctx = core.Context(0)
    ctx.set_option("accel", 0)
    ds = core.DeviceScene(cb, ctx=ctx)
    ref = ds.render(nx, ny, ns)
    assert ctx.last_accel() == "flat" and np.array_equal(ref[0], out[None][0])
    ds.close(); ctx.close()
    # not small, or not mixed-kind: the tree
    for scene in (r.scene.make_final(64, 64), r.scene.make_two_spheres(64, 64), r.scene.make_random_scene(64, 64, 1, False)):
        ctx = core.Context(0)
        ds = core.DeviceScene(fl.flatten(scene), ctx=ctx)
        with pytest.raises(core.RtmiError):
            ctx.last_accel()  # nothing rendered yet
        ds.render(64, 64, 2)
        assert ctx.last_accel() == "bvh"
        ds.close(); ctx.close()
    ctx = core.Context(0)
    with pytest.raises(core.RtmiError):
        ctx.set_option("flat_below", -1)
    ctx.close()


# ---- the traversal's float bound of the closest hit (rtmi_device.h: float_above) -----------------------------------------------------------------------------
def test_float_bound_of_the_closest_hit_is_a_bound():
    """the BVH traversal prunes boxes against a FLOAT upper bound of the FP64 closest hit, refreshed after every exact-test phase; since round 3 it
    is RN(t)(1 + 2^-23) + 2^-120 (three instructions) instead of the exact next float above t: it must never be below t, whatever t -- positive,
    negative (probes with t-min < 0), denormal, a float already, just above / below a float, huge -- and it should stay within two float ulps"""
    rng = np.random.default_rng(3)
    f = rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 20000).astype(np.float32)
    f = f[np.isfinite(f)]
    exact = f.astype(np.float64)
    t = np.concatenate([exact, np.nextafter(exact, np.inf), np.nextafter(exact, -np.inf), exact * (1 + 2.0 ** -30), exact * (1 - 2.0 ** -30),
                        10.0 ** rng.uniform(-300, 300, 20000) * rng.choice([-1.0, 1.0], 20000), [0.0, -0.0, 5e-324, -5e-324, 1e-46, 1.4e-45, 3.4028234663852886e38,
                                                                                                 3.4028235e38, 1e39, -1e39, 1e300, 0.001, 1.0]])
    out = core.probe_math(np.stack([np.ones_like(t), np.ones_like(t), t], axis=1))[:, 8]
    fmax = 3.4028234663852886e38
    assert np.all(np.isfinite(out)) and np.all(out == out.astype(np.float32).astype(np.float64)), "a float"
    inside = np.abs(t) < fmax
    assert np.all(out[inside] >= t[inside]), "never below t"
    assert np.all(out[t >= fmax] == fmax), "beyond the float range: FLT_MAX (the closest hit so far never exceeds the caller's t-max = Float/MAX_VALUE)"
    normal = inside & (np.abs(t) > 1e-25)  # (below, the absolute term 2^-120 that covers the denormal range shows)
    up = np.nextafter(np.nextafter(np.nextafter(t[normal].astype(np.float32), np.float32(np.inf)), np.float32(np.inf)), np.float32(np.inf)).astype(np.float64)
    assert np.all(out[normal] <= up), "within a few float ulps"

"""GPU tests added in round 2: the torch-free host, the single-process multi-device entry, the traversal counters, the
region render, Isotropic on a plain sphere, stream ordering, and BASELINE's full-size configurations C3 / C4 / C5 through
size-independent properties plus oracle spot regions at full sample counts.  Everything goes through the C-ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import raytrace_clj_amd as r
from raytrace_clj_amd import core
from raytrace_clj_amd import flatten as fl
from raytrace_clj_amd.util import vec3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
RMS_TOL = 1e-4


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


# ---- the reference-side binding's situation: a host without torch (and without Python) ---------------------------------
def test_torch_free_c_host_renders_the_golden_fixture(tmp_path):
    """tests/host_smoke.c dlopen()s librtmi.so in a process that has never seen torch (the library then binds /opt/rocm's HIP
    runtime, as under the JVM), creates the render_cover_n3 scene through rtmi_scene_create, calls rtmi_render, then clones the
    scene and calls rtmi_render_multi with two replicas; both frames must equal the committed fixture."""
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
    out = subprocess.run([str(exe), r._ffi.LIB_PATH, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "backend=hip-gfx950" in out.stdout and "arch=gfx950" in out.stdout
    raw = open(tmp_path / "out.bin", "rb").read()
    n = nx * ny * 3
    one = n * 8 + n + 16
    assert len(raw) == 2 * one
    for k in range(2):  # rtmi_render, then rtmi_render_multi over two replicas
        blob = raw[k * one:(k + 1) * one]
        lin = np.frombuffer(blob[:n * 8], np.float64).reshape(ny, nx, 3)
        q = np.frombuffer(blob[n * 8:n * 8 + n], np.uint8).reshape(ny, nx, 3)
        cnt = np.frombuffer(blob[n * 8 + n:], np.uint64)
        assert rms(lin, z["linear"]) < 1e-13 and np.array_equal(cnt, z["counters"]), k
        assert np.abs(q.astype(int) - z["rgb8"].astype(int)).max() <= 1


# ---- one host process, several replicas -----------------------------------------------------------------------------------
def test_multi_device_entry_matches_single_device(cover11_moving):
    """rtmi_render_multi / rtmi_render_multi_device with 3 replicas sharing this box's GPU (device-copy gather): tile dealing,
    per-replica renders on their own streams, gather, assemble, counter sums -- the image is bit-identical to rtmi_render"""
    import torch
    from raytrace_clj_amd import dist as rdist
    nx, ny, ns = 200, 100, 8
    flat = fl.flatten(cover11_moving)
    ds = core.DeviceScene(flat)
    lin, q, cnt = ds.render(nx, ny, ns)
    ds.close()
    md = rdist.MultiDevice(flat, [0, 0, 0])
    mlin, mq, mcnt = md.render(nx, ny, ns)
    assert np.array_equal(mlin, lin) and np.array_equal(mq, q) and np.array_equal(mcnt, cnt)
    dl = torch.zeros((ny, nx, 3), dtype=torch.float64, device="cuda")
    dq = torch.zeros((ny, nx, 3), dtype=torch.uint8, device="cuda")
    dc = torch.zeros(2, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):  # back-to-back frames on the same replicas
        md.render_device(nx, ny, ns, dl, dq, dc)
    md.sync()
    assert np.array_equal(dl.cpu().numpy(), lin) and np.array_equal(dq.cpu().numpy(), q) and [int(dc[0]), int(dc[1])] == [int(cnt[0]), int(cnt[1])]
    assert md.last_gather_ms() >= 0.0
    with pytest.raises(core.RtmiError):  # two replicas on one context: a context is not re-entrant
        import ctypes as C
        arr = (C.c_void_p * 2)(md.scenes[0].handle, md.scenes[0].handle)
        core.check(r._ffi.lib().rtmi_render_multi(2, arr, nx, ny, ns, 50, 1, 0, None, None, None))
    md.close()
    one = rdist.MultiDevice(flat, [0])  # n = 1: no gather at all
    olin, oq, ocnt = one.render(nx, ny, ns)
    one.close()
    assert np.array_equal(olin, lin) and np.array_equal(oq, q) and np.array_equal(ocnt, cnt)
    # replicas must be clones of one scene
    ctx_a, ctx_b = core.Context(0), core.Context(0)
    sa, sb = core.DeviceScene(flat, ctx=ctx_a), core.DeviceScene(fl.flatten(r.scene.make_random_scene(200, 100, 3, False)), ctx=ctx_b)
    import ctypes as C
    arr = (C.c_void_p * 2)(sa.handle, sb.handle)
    with pytest.raises(core.RtmiError) as e:
        core.check(r._ffi.lib().rtmi_render_multi(2, arr, nx, ny, ns, 50, 1, 0, None, None, None))
    assert e.value.code == -1
    sb.close(); sa.close(); ctx_b.close(); ctx_a.close()


def test_multi_device_entry_on_distinct_devices(cover11):
    """the RCCL path proper (ncclCommInitAll + ONE ncclGather inside the library): needs two visible devices"""
    import torch
    from raytrace_clj_amd import dist as rdist
    n = min(2, torch.cuda.device_count())
    if n < 2:
        pytest.skip("one visible device: the in-library RCCL gather needs two (the shared-device form is tested above)")
    nx, ny, ns = 200, 100, 8
    flat = fl.flatten(cover11)
    ds = core.DeviceScene(flat)
    lin, q, cnt = ds.render(nx, ny, ns)
    ds.close()
    md = rdist.MultiDevice(flat, list(range(n)))
    for _ in range(2):
        mlin, mq, mcnt = md.render(nx, ny, ns)
        assert np.array_equal(mlin, lin) and np.array_equal(mq, q) and np.array_equal(mcnt, cnt)
    md.close()


def test_scene_clone_carries_perlin_images_and_media():
    """rtmi_scene_clone of make-final (Perlin tables, an ImageMap, two ConstantMedium with their call sequence, instances)"""
    nx, ny, ns = 48, 48, 4
    sc = r.scene.make_final(nx, ny)
    ctx1, ctx2 = core.Context(0), core.Context(0)
    ds = core.DeviceScene(sc, ctx=ctx1)
    cl = ds.clone(ctx2)
    a, b = ds.render(nx, ny, ns), cl.render(nx, ny, ns)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[0].mean() > 0.01
    cl.close(); ds.close(); ctx2.close(); ctx1.close()


# ---- metrics.clj:10 aabb.intersection.total ------------------------------------------------------------------------------
def test_traversal_counters(cover11):
    nx, ny, ns = 200, 100, 8
    ctx = core.Context(0)
    ds = core.DeviceScene(cover11, ctx=ctx)
    base = ds.render(nx, ny, ns)
    with pytest.raises(core.RtmiError):
        ctx.last_traversal_counters()  # not enabled
    ctx.set_option("count_traversal", 1)
    counted = ds.render(nx, ny, ns)
    aabb, prims = ctx.last_traversal_counters()
    again = ds.render(nx, ny, ns)
    assert ctx.last_traversal_counters() == (aabb, prims), "deterministic"
    for x, y, w in zip(base, counted, again):
        assert np.array_equal(x, y) and np.array_equal(x, w), "the counting instantiation renders the same image"
    rays = int(base[2][0])
    assert aabb % 2 == 0 and 2 * rays < aabb < 200 * rays, (aabb, rays)       # a few to a few dozen node visits per segment
    assert 2 * rays <= prims < 40 * rays, (prims, rays)                         # dome + ground for every ray, plus leaves
    ctx.set_option("workspace_bytes", 4 << 20)                                  # several sample passes: the counters accumulate over them
    ds.render(nx, ny, ns)
    assert ctx.last_traversal_counters() == (aabb, prims)
    ds.close(); ctx.close()


# ---- rtmi_render on a region ------------------------------------------------------------------------------------------------
def test_region_render_is_the_crop_and_counts_the_region(oracle, cover_small):
    nx, ny, ns = 61, 37, 5  # partial tiles on both edges
    f = fl.flatten(cover_small)
    ds = core.DeviceScene(f)
    full, q, cnt = ds.render(nx, ny, ns)
    for region in [(0, 0, nx, ny), (3, 5, 29, 30), (56, 32, 61, 37), (8, 8, 16, 16), (60, 36, 61, 37), (0, 0, 1, 37)]:
        x0, y0, x1, y1 = region
        part, qp, cp = ds.render(nx, ny, ns, region=region)
        assert np.array_equal(part, full[y0:y1, x0:x1]) and np.array_equal(qp, q[y0:y1, x0:x1]), region
        _, _, ecnt = oracle.render(f, nx, ny, ns, 50, core.RENDER_SEED, region=region, nthreads=8)
        assert np.array_equal(cp, ecnt), (region, cp, ecnt)
    ds.close()


# ---- ADVICE: Isotropic on a primitive of a sphere-only scene ---------------------------------------------------------------
def test_isotropic_material_on_a_plain_sphere_scatters(oracle):
    """shader.clj:129-138: Isotropic.scatter always scatters; a plain Sphere carrying it must not render as an absorber"""
    iso = r.shader.isotropic(albedo=r.texture.constant(color=vec3(0.7, 0.6, 0.5)))
    world = r.hitable.hitlist(items=[r.hitable.sphere(center=vec3(0, 0, 0), radius=1000, material=r.shader.diffuse_light(tex=r.texture.constant(color=vec3(1, 1, 1)))),
                                     r.hitable.sphere(center=vec3(0, 0, -3), radius=1.0, material=iso)])
    cam = r.camera.pinhole_camera(lookfrom=vec3(0, 0, 2), lookat=vec3(0, 0, -3), vup=vec3(0, 1, 0), vfov=40.0, aspect=1.0)
    f = fl.flatten({"camera": cam, "world": world})
    ds = core.DeviceScene(f)
    rng = np.random.default_rng(4)
    n = 2000
    rays = np.concatenate([np.tile(vec3(0, 0, 2), (n, 1)), rng.normal(0, 1, (n, 3)), rng.random((n, 1))], axis=1)
    hits = np.concatenate([rng.normal(0, 1, (n, 3)), np.tile(vec3(0, 0, 1), (n, 1)), rng.random((n, 2))], axis=1)
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    mat = int(np.flatnonzero(f.mat_kind == 4)[0])
    got, exp = ds.probe_scatter(mat, rays, hits, keys), oracle.probe_scatter(f, mat, rays, hits, keys)
    assert (got[:, 0] == 1).all() and np.array_equal(got[:, :7], exp[:, :7]) and np.array_equal(got[:, 8], exp[:, 8])
    lin, q, cnt = ds.render(32, 32, 8)
    elin, eq, ecnt = oracle.render(f, 32, 32, 8, 50, core.RENDER_SEED, nthreads=8)
    ds.close()
    assert np.array_equal(cnt, ecnt) and rms(lin, elin) < 1e-13
    assert lin[12:20, 12:20].mean() > 0.3, "the ball scatters light (it is not black)"


# ---- ADVICE: ordering with torch's default stream ----------------------------------------------------------------------------
def test_tile_renderer_is_ordered_with_the_default_stream(cover11):
    """TileRenderer.step() from torch's default stream (handle 0 = the C-ABI's "context stream"): the render must start after
    the torch work queued before it and the torch work queued after it must see the finished frame"""
    import torch
    from raytrace_clj_amd import dist as rdist
    nx, ny, ns = 400, 200, 16
    ctx = core.Context(0)
    ds = core.DeviceScene(cover11, ctx=ctx)
    ref = ds.render(nx, ny, ns)
    tr = rdist.TileRenderer(ds, nx, ny, 0, 1)
    assert torch.cuda.current_stream().cuda_stream == 0
    for _ in range(3):
        big = torch.randn(4096, 4096, device="cuda")
        for _ in range(10):
            big = big @ big * 1e-3           # keeps the default stream busy for a while
        tr.local.fill_(float("nan"))         # queued on the default stream BEFORE the render: must not land after it
        tr.linear.fill_(float("nan"))
        tr.step(ns)
        snap = tr.linear.clone()             # queued on the default stream AFTER the render: must see the finished frame
        torch.cuda.synchronize()
        assert np.array_equal(snap.cpu().numpy(), ref[0])
    ds.close(); ctx.close()


# ---- BASELINE full sizes ---------------------------------------------------------------------------------------------------
def _spot(oracle, f, ds, nx, ny, ns, region, threads=64):
    lin, q, cnt = ds.render(nx, ny, ns, region=region)
    exp, eq, ecnt = oracle.render(f, nx, ny, ns, 50, core.RENDER_SEED, region=region, nthreads=threads)
    assert rms(lin, exp) <= RMS_TOL and rms(lin, exp) < 1e-13, (region, rms(lin, exp))
    assert np.array_equal(cnt, ecnt) and np.abs(q.astype(int) - eq.astype(int)).max() <= 1
    return lin


def test_config_c3_full_size(oracle):
    """BASELINE configs[2], the north star's configuration: 1920x1080x256spp, cover scene n=50 (10 003 spheres).  Full frame at
    full spp: determinism, sample-pass split invariance (1 pass by default; 13 with a 1 GiB workspace), counters in range; one
    16x8 region at full spp against the oracle (bit-level agreement), which must also be the crop of the full frame."""
    nx, ny, ns = 1920, 1080, 256
    sc = r.scene.make_random_scene(nx, ny, 50, False)
    f = fl.flatten(sc)
    assert f.n_prims > 9900
    ctx = core.Context(0)
    ds = core.DeviceScene(f, ctx=ctx)
    base, q, cnt = ds.render(nx, ny, ns)
    again, q2, cnt2 = ds.render(nx, ny, ns)
    assert np.array_equal(base, again) and np.array_equal(q, q2) and np.array_equal(cnt, cnt2), "deterministic"
    assert cnt[1] == nx * ny and 1.5 * nx * ny * ns < cnt[0] < 6 * nx * ny * ns
    ctx.set_option("workspace_bytes", 1 << 30)
    split, _, cnt3 = ds.render(nx, ny, ns)
    ctx.set_option("workspace_bytes", 16 << 30)
    assert np.array_equal(base, split) and np.array_equal(cnt, cnt3), "sample-pass split must not change the image"
    region = (952, 620, 968, 628)
    lin = _spot(oracle, f, ds, nx, ny, ns, region)
    assert np.array_equal(lin, base[620:628, 952:968])
    ds.close(); ctx.close()
    assert np.isfinite(base).all() and base[:40].mean() > 0.5 and base.min() >= 0


def test_config_c5_full_size(oracle):
    """BASELINE configs[4]: 1920x1080x4096spp, dielectric-heavy cover scene (80 % glass), 3 sample passes through the default
    64 GiB sample buffer (the many-pass form, 24+ passes through 8 GiB, is tests/test_gpu_round3.py::test_config_c5_many_passes):
    determinism, counters in range (long specular chains), one oracle spot region at the full 4096 spp that is also the crop of the frame."""
    nx, ny, ns = 1920, 1080, 4096
    sc = r.scene.make_random_scene(nx, ny, 11, False, mix=(0.1, 0.2))
    f = fl.flatten(sc)
    ctx = core.Context(0)
    ds = core.DeviceScene(f, ctx=ctx)
    base, q, cnt = ds.render(nx, ny, ns)
    again, q2, cnt2 = ds.render(nx, ny, ns)
    assert np.array_equal(base, again) and np.array_equal(q, q2) and np.array_equal(cnt, cnt2), "deterministic"
    assert cnt[1] == nx * ny and 2.5 * nx * ny * ns < cnt[0] < 8 * nx * ny * ns
    region = (1000, 700, 1008, 704)  # 32 pixels x 4096 spp on the glass-sphere field
    lin = _spot(oracle, f, ds, nx, ny, ns, region)
    assert np.array_equal(lin, base[700:704, 1000:1008])
    ds.close(); ctx.close()
    assert np.isfinite(base).all() and base.min() >= 0


def test_config_c4_frame_partitioned_over_8_ranks():
    """BASELINE configs[3]'s frame (3840x2160, cover scene n=11) cut over 8 ranks exactly as the 8-GPU run cuts it (tiles r, r+8,
    ...; 129 600 tiles, 16 200 per rank), the ranks rendered one after the other on this GPU at 4 spp, then assembled: equal
    to the un-partitioned render, counters add up.  Then the same through the single-process entry with 8 replicas."""
    import torch
    from raytrace_clj_amd import dist as rdist
    nx, ny, ns = 3840, 2160, 4
    flat = fl.flatten(r.scene.make_random_scene(nx, ny, 11, False))
    ctx = core.Context(0)
    ds = core.DeviceScene(flat, ctx=ctx)
    base, q, cnt = ds.render(nx, ny, ns)
    L = r._ffi.lib()
    world = 8
    per = int(L.rtmi_local_tiles(nx, ny, 0, world))
    assert per == 16200
    gathered = torch.zeros((world, per, 64, 3), dtype=torch.float64, device="cuda")
    counters = torch.zeros((world, 2), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for rank in range(world):
        ds.render_tiles_device(nx, ny, ns, rank, world, gathered[rank], counters[rank])
    out = torch.zeros((ny, nx, 3), dtype=torch.float64, device="cuda")
    out8 = torch.zeros((ny, nx, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    core.check(L.rtmi_assemble_device(ctx.handle, nx, ny, world, per, r._ffi.ptr(gathered), r._ffi.ptr(out), r._ffi.ptr(out8), None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), base) and np.array_equal(out8.cpu().numpy(), q), "tile partition invariance at 3840x2160"
    assert int(counters[:, 0].sum()) == int(cnt[0]) and int(counters[:, 1].sum()) == nx * ny
    balance = counters[:, 0].double() / counters[:, 0].double().mean()
    assert float(balance.min()) > 0.97 and float(balance.max()) < 1.03, "round-robin 8x8 tiles balance the ray segments across ranks"
    ds.close(); ctx.close()
    del gathered, out, out8
    md = rdist.MultiDevice(flat, [0] * 8)
    mlin, mq, mcnt = md.render(nx, ny, ns)
    md.close()
    assert np.array_equal(mlin, base) and np.array_equal(mq, q) and np.array_equal(mcnt, cnt)


def test_time_sliced_traversal_does_not_change_the_image():
    """rtmi option "suspend_lanes": the threshold below which a wave's BVH traversal parks its last lanes (their cursor, stack column
    and closest hit so far) and hands the wave back; per lane the visiting order is the same depth-first order, so every threshold --
    0 = the plain while-while loop, 64 = hand back after every exact-test phase -- must give the same bits, counters included, in
    both precisions and with moving spheres (whose out-of-shutter fallback runs after a resumed traversal too)"""
    for moving, precision in ((False, "f64"), (True, "f64"), (False, "f32")):
        scene = r.scene.make_random_scene(160, 80, 11, moving, mix=(0.6, 0.85))
        flat = fl.flatten(scene)
        ref = None
        for lanes in (0, 1, 8, 33, 64):
            ctx = core.Context(0)
            ctx.set_option("suspend_lanes", lanes)
            ctx.set_option("count_traversal", 1)
            ds = core.DeviceScene(flat, ctx=ctx)
            lin, q, cnt = ds.render(160, 80, 24, precision=precision)
            trav = ctx.last_traversal_counters()
            ds.close(); ctx.close()
            if ref is None:
                ref = (lin, q, cnt, trav)
            else:
                assert np.array_equal(lin, ref[0]) and np.array_equal(q, ref[1]) and list(cnt) == list(ref[2]) and trav == ref[3], (moving, precision, lanes)
    with pytest.raises(core.RtmiError):
        core.Context(0).set_option("suspend_lanes", 65)


def test_reduce_timing_window():
    """rtmi_last_reduce_ms reports the in-order sample reductions of the window rtmi_last_trace_ms closed: one per trace launch (several when
    the sample buffer forces passes), each a positive duration well below the trace kernel's"""
    scene = r.scene.make_random_scene(320, 160, 11, False)
    ctx = core.Context(0, timing=True)
    ds = core.DeviceScene(fl.flatten(scene), ctx=ctx)
    ds.render(320, 160, 32)
    t_ms, t_n = ctx.last_trace_ms()
    r_ms, r_n = ctx.last_reduce_ms()
    assert t_n == 1 and r_n == 1 and 0.0 < r_ms < t_ms
    ctx.set_option("workspace_bytes", 4 << 20)  # 320 x 160 x 24 B = 1.2 MB per sample: 3 samples per pass
    ds.render(320, 160, 32)
    t_ms, t_n = ctx.last_trace_ms()
    r_ms, r_n = ctx.last_reduce_ms()
    assert t_n == r_n == 11 and 0.0 < r_ms < t_ms
    ds.close(); ctx.close()
    plain = core.Context(0)
    with pytest.raises(core.RtmiError):
        plain.last_reduce_ms()
    plain.close()

"""CPU tests of the host logic: the reference-protocol mirror, the flattener, the scene generator, the
golden fixtures against the oracle, and that librtmi.so loads and exports every symbol include/rtmi.h declares
(no compute call is made here: there is no GPU)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import raytrace_clj_amd as r
from raytrace_clj_amd import _ffi
from raytrace_clj_amd import flatten as fl
from raytrace_clj_amd.util import vec3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


class _Flat:
    def __init__(self, z):
        for k in ("prim_kind", "prim_geom", "prim_mat", "mat_kind", "mat_tex", "mat_param", "tex_kind", "tex_param", "tex_child", "cam"):
            setattr(self, k, z[k])
        self.cam_kind = int(z["cam_kind"])


# ---- util_test.clj:7-41 (vec3, ray) on the host mirror ---------------------------------------------------------
def test_vec3_and_ray():
    for v in (vec3(1, 2, 3), vec3(1.0, 2.0, 3.0)):
        assert v.dtype == np.float64 and list(v) == [1.0, 2.0, 3.0]
    from fractions import Fraction
    assert list(vec3(Fraction(1, 2), Fraction(1, 4), Fraction(1, 8))) == [0.5, 0.25, 0.125]
    with pytest.raises(IndexError):
        vec3(1, 2, 3)[3]
    rr = r.util.ray(vec3(1, 2, 3), vec3(4, 5, 6), 0.1)
    assert set(rr) == {"origin", "direction", "time"} and rr["time"] == 0.1
    assert list(r.util.point_at_parameter(rr, 1)) == [5, 7, 9] and list(r.util.point_at_parameter(rr, -1)) == [-3, -3, -3]


def test_record_field_names_match_reference():
    import dataclasses as dc
    names = lambda c: [f.name for f in dc.fields(c)]
    assert names(r.hitable.Hitlist) == ["items"] and names(r.hitable.bvh_node) == ["left", "right", "box"]
    assert names(r.hitable.Sphere) == names(r.hitable.UVSphere) == ["center", "radius", "material"]
    assert names(r.hitable.MovingSphere) == ["center0", "t0", "center1", "t1", "radius", "material"]
    assert names(r.hitable.AABB) == ["vmin", "vmax"]
    assert names(r.shader.Lambertian) == ["albedo"] and names(r.shader.Metal) == ["albedo", "fuzz"]
    assert names(r.shader.Dielectric) == ["ri"] and names(r.shader.DiffuseLight) == ["tex"]
    assert names(r.texture.Constant) == ["color"] and names(r.texture.Checkerboard) == ["tex0", "tex1", "scale"]
    assert names(r.texture.UVGradient) == ["co", "cu", "cv", "cuv"]
    assert names(r.camera.PinholeCamera) == ["origin", "lleft", "horiz", "vert"]
    assert names(r.camera.ThinLensCamera) == ["origin", "lleft", "horiz", "vert", "u", "v", "w", "aperture", "t0", "t1"]


def test_make_bvh_shapes():
    m = r.shader.dielectric(ri=1.5)
    pts = [vec3(0, 0, 0), vec3(1, 2, 3), vec3(-2, -1, -3), vec3(3, -1, 2), vec3(-3, 2, 1)]  # hitable_test.clj:150-156
    spheres = [r.hitable.sphere(center=p, radius=0.1, material=m) for p in pts]
    one = r.hitable.make_bvh(spheres[:1], 0, 1)
    assert one.left is one.right is spheres[0]  # hitable.clj:113-114
    tree = r.hitable.make_bvh(spheres, 0, 1)
    assert isinstance(tree, r.hitable.bvh_node)
    assert np.array_equal(tree.box.vmin, [-3.1, -1.1, -3.1]) and np.array_equal(tree.box.vmax, [3.1, 2.1, 3.1])
    f = fl.flatten(tree, r.camera.pinhole_camera(lookfrom=vec3(0, 0, 5), lookat=vec3(0, 0, 0), vup=vec3(0, 1, 0), vfov=40, aspect=2))
    assert f.n_prims == 5 and len(f.mat_kind) == 1  # leaves de-duplicated, material interned
    assert fl.flatten(one, None).n_prims == 1


def test_flatten_cover_scene(cover11, cover11_moving):
    f = fl.flatten(cover11)
    assert 485 <= f.n_prims <= 489  # 484 candidates minus those near (4,.2,0) plus 5 hero
    assert (f.prim_kind == fl.PRIM_UVSPHERE).sum() == 1 and (f.prim_kind == fl.PRIM_MOVING).sum() == 0
    kinds = f.mat_kind[f.prim_mat]
    small = f.prim_geom[:, 3] == 0.2
    frac = [(kinds[small] == k).mean() for k in (fl.MAT_LAMBERTIAN, fl.MAT_METAL, fl.MAT_DIELECTRIC)]
    assert 0.7 < frac[0] < 0.9 and 0.08 < frac[1] < 0.22 and 0.01 < frac[2] < 0.1
    assert (f.tex_kind == fl.TEX_CHECKER).sum() == 1 and (f.tex_kind == fl.TEX_UVGRADIENT).sum() == 1
    ck = int(np.flatnonzero(f.tex_kind == fl.TEX_CHECKER)[0])
    assert f.tex_param[ck, 0] == 10 and (f.tex_child[ck] < ck).all()
    d = np.linalg.norm(f.prim_geom[small, 0:3] - vec3(4, 0.2, 0), axis=1)
    assert (d > 0.9).all()  # scene.clj:375
    fm = fl.flatten(cover11_moving)
    mv = fm.prim_kind == fl.PRIM_MOVING
    assert mv.sum() > 300 and (fm.prim_geom[mv, 7] == 0).all() and (fm.prim_geom[mv, 8] == 1).all()
    dy = fm.prim_geom[mv, 5] - fm.prim_geom[mv, 1]
    assert (dy >= 0).all() and (dy < 0.5).all() and (fm.prim_geom[mv, 4] == fm.prim_geom[mv, 0]).all()
    # seeded: same seed -> same scene, another seed -> another scene; a bvh world and a hitlist world hold the same leaves
    again = fl.flatten(r.scene.make_random_scene(200, 100, 11, False))
    assert np.array_equal(np.sort(again.prim_geom, axis=0), np.sort(f.prim_geom, axis=0))
    other = fl.flatten(r.scene.make_random_scene(200, 100, 11, False, seed=99))
    assert not np.array_equal(np.sort(other.prim_geom, axis=0), np.sort(f.prim_geom, axis=0))
    hl = fl.flatten(r.scene.make_random_scene(200, 100, 11, False, bvh=False))
    assert hl.n_prims == f.n_prims and hl.prim_kind[0] == fl.PRIM_UVSPHERE and hl.prim_geom[1, 1] == -1000
    n50 = fl.flatten(r.scene.make_random_scene(1920, 1080, 50, False))
    assert 9990 <= n50.n_prims <= 10005
    heavy = fl.flatten(r.scene.make_random_scene(200, 100, 11, False, mix=(0.1, 0.2)))
    hk = heavy.mat_kind[heavy.prim_mat][heavy.prim_geom[:, 3] == 0.2]
    assert (hk == fl.MAT_DIELECTRIC).mean() > 0.7


def test_flatten_rejects_unknown_records():
    class RectXY:
        pass
    with pytest.raises(r.UnsupportedOnGpuPath):
        fl.flatten(r.hitable.hitlist(items=[RectXY()]), None)
    class Isotropic(r.shader.Shader):
        pass
    with pytest.raises(r.UnsupportedOnGpuPath):
        fl.flatten(r.hitable.hitlist(items=[r.hitable.sphere(center=vec3(0, 0, 0), radius=1, material=Isotropic())]), None)
    class Marble(r.texture.Texture):
        pass
    with pytest.raises(r.UnsupportedOnGpuPath):
        fl.flatten(r.hitable.hitlist(items=[r.hitable.sphere(center=vec3(0, 0, 0), radius=1, material=r.shader.lambertian(albedo=Marble()))]), None)


# ---- golden fixtures: the oracle reproduces them (so a changed oracle cannot silently move the goalposts) ------------
def test_flatten_media_modes():
    """where a ConstantMedium stands decides the t-max its hit? is handed: below bvh-nodes only -> RTMI_MEDIA_DESCENT (un-narrowed,
    hitable.clj:99-105), inside Hitlists only -> RTMI_MEDIA_HITLIST (narrowed by the items before it, hitable.clj:15-26: list order kept,
    nested Hitlists and Boxes spliced in), both at once -> unsupported"""
    H, S, T = r.hitable, r.shader, r.texture
    grey = S.lambertian(albedo=T.constant(color=vec3(0.5, 0.5, 0.5)))
    a, b, c = (H.sphere(center=vec3(k, 0, 0), radius=0.4, material=grey) for k in range(3))
    fog = H.constant_medium(boundary=H.sphere(center=vec3(1, 0, 0), radius=2.0, material=grey), density=0.5, albedo=T.constant(color=vec3(1, 1, 1)))
    cam = r.camera.pinhole_camera(lookfrom=vec3(0, 0, 5), lookat=vec3(0, 0, 0), vup=vec3(0, 1, 0), vfov=40.0, aspect=1.0)
    from raytrace_clj_amd import flatten
    f = flatten.flatten({"camera": cam, "world": H.hitlist(items=[a, H.hitlist(items=[b, fog]), c])})
    assert f.media_mode == 1 and list(f.media_calls) == [2] and f.n_prims == 4 and list(f.prim_kind[:4] & 15) == [0, 0, 7, 0]
    f = flatten.flatten({"camera": cam, "world": H.make_bvh([a, b, fog, c], 0.0, 1.0)})
    assert f.media_mode == 0 and len(f.media_calls) >= 1
    # a Hitlist holding a medium BELOW a bvh-node (rejected until round 3): narrowed per call -- by the items before it in its own list, not by the bvh siblings
    f = flatten.flatten({"camera": cam, "world": H.bvh_node(a, H.hitlist(items=[b, fog, c]), a.bbox(0.0, 1.0))})  # (the reference's Hitlist has no bbox: built by hand)
    assert f.media_mode == 2 and list(f.media_calls) == [2] and list(f.media_narrow_from) == [1] and list(f.prim_kind[:4] & 15) == [0, 0, 7, 0]
    f = flatten.flatten({"camera": cam, "world": H.bvh_node(fog, H.hitlist(items=[b, H.hitlist(items=[c, fog])]), a.bbox(0.0, 1.0))})  # the same record: un-narrowed in the tree, narrowed in the (nested) list
    assert f.media_mode == 2 and list(f.media_calls) == [0, 3] and list(f.media_narrow_from) == [0, 1]
    with pytest.raises(flatten.UnsupportedOnGpuPath):  # a bvh-node INSIDE a Hitlist above the medium: narrowed by the list's earlier items but not by its bvh siblings
        flatten.flatten({"camera": cam, "world": H.hitlist(items=[a, H.bvh_node(b, fog, a.bbox(0.0, 1.0))])})
    with pytest.raises(flatten.UnsupportedOnGpuPath):  # the narrowing list shares a record with another part of the world: its items are not contiguous
        flatten.flatten({"camera": cam, "world": H.bvh_node(a, H.hitlist(items=[a, fog]), a.bbox(0.0, 1.0))})
    # the same record listed twice in a Hitlist (rejected until round 3): every listing is a primitive of its own, at its own place
    f = flatten.flatten({"camera": cam, "world": H.hitlist(items=[a, fog, b, fog])})
    assert f.media_mode == 1 and list(f.media_calls) == [1, 3] and f.n_prims == 4 and list(f.prim_kind[:4] & 15) == [0, 7, 0, 7]
    assert list(f.prim_geom[1][:1]) == list(f.prim_geom[3][:1]) and f.prim_geom[1][1] != f.prim_geom[3][1]  # same density, each listing its own copy of the boundary


def test_golden_rng():
    cases = json.load(open(os.path.join(GOLD, "rng.json")))["cases"]
    from oracle.oracle import Oracle
    o64, o32 = Oracle("f64"), Oracle("f32")
    for c in cases:
        k = int(c["key"])
        assert o64.sample_key(int(c["seed"]), int(c["pixel"]), int(c["sample"])) == k
        for d, (b, u53, u24) in enumerate(zip(c["bits"], c["u53"], c["u24"])):
            assert o64.draw_bits(k, d) == int(b) and o64.draw(k, d) == u53 and o32.draw(k, d) == u24


@pytest.mark.parametrize("name", ["render_cover_n3.npz", "render_two_spheres.npz"])
def test_golden_renders(oracle, name):
    z = np.load(os.path.join(GOLD, name))
    lin, q, cnt = oracle.render(_Flat(z), int(z["nx"]), int(z["ny"]), int(z["ns"]), int(z["depth"]), int(z["seed"]), nthreads=4)
    assert np.array_equal(lin, z["linear"]) and np.array_equal(q, z["rgb8"]) and np.array_equal(cnt, z["counters"])


def test_golden_cornell(oracle):
    """the committed Cornell fixture is reproduced by the nested oracle, and the flattener still produces the stored arrays"""
    from oracle.tree import flatten_with_tree
    z = np.load(os.path.join(GOLD, "render_cornell.npz"))
    f = flatten_with_tree(r.scene.make_cornell_box(40, 40))
    for k in ("prim_kind", "prim_geom", "prim_mat", "prim_flip", "prim_xform", "xform_kind", "xform_param", "cam"):
        assert np.array_equal(getattr(f, k), z[k]), k
    lin, q, cnt = oracle.render(f, 40, 40, 8, 50, int(z["seed"]), nthreads=4)
    assert np.array_equal(lin, z["linear"]) and np.array_equal(q, z["rgb8"]) and np.array_equal(cnt, z["counters"])


def test_golden_scene_generator_is_stable():
    z = np.load(os.path.join(GOLD, "render_cover_n3.npz"))
    f = fl.flatten(r.scene.make_random_scene(48, 24, 3, False))
    assert np.array_equal(f.prim_geom, z["prim_geom"]) and np.array_equal(f.cam, z["cam"]) and np.array_equal(f.tex_param, z["tex_param"])


def test_golden_paths(oracle):
    z = np.load(os.path.join(GOLD, "paths_cover11_moving.npz"))
    rgb, nseg, log, nlog = oracle.probe_paths(_Flat(z), z["rays"], z["keys"], int(z["depth"]), int(z["ctr0"]), max_seg=8)
    assert np.array_equal(rgb, z["rgb"]) and np.array_equal(nseg, z["nseg"]) and np.array_equal(log, z["log"])
    assert nseg.max() > 3 and (z["log"][:, :, 0].max() > 5)


def test_f32_oracle_close_to_f64(oracle, oracle_f32, cover_small):
    f = fl.flatten(cover_small)
    a, _, _ = oracle.render(f, 24, 12, 16, seed=5)
    b, _, _ = oracle_f32.render(f, 24, 12, 16, seed=5)
    assert np.sqrt(np.mean((a - b) ** 2)) < 0.08  # different uniforms (24 vs 53 bits) and chaotic paths: statistical only


# ---- the C-ABI library: loads, and exports exactly what include/rtmi.h declares ---------------------------------
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_ffi.LIB_PATH), "build with `make -C raytrace_clj_amd/csrc` or __graft_entry__.build()"
    L = ctypes.CDLL(_ffi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(_ffi.SYMBOLS) == declared  # the python binding covers the whole header
    L.rtmi_backend_name.restype = ctypes.c_char_p
    assert L.rtmi_backend_name() == b"hip-gfx950"
    L.rtmi_sample_key.restype = ctypes.c_uint64
    L.rtmi_sample_key.argtypes = [ctypes.c_uint64] * 3
    assert L.rtmi_sample_key(0x5EED0002, 319999, 63) == r.util.sample_key(0x5EED0002, 319999, 63)  # pure host function
    assert L.rtmi_local_tiles(800, 400, 0, 1) == 5000 and L.rtmi_local_tiles(800, 400, 3, 8) == 625
    assert L.rtmi_local_tiles(200, 100, 0, 1) == 25 * 13


def test_rccl_probe_reports_a_missing_library_without_crashing():
    """rtmi_rccl_probe: the open / resolve step of the in-library RCCL gather, callable without a device.  A missing library must come back
    as RTMI_E_DEVICE with the loader's message (round 2 built that message from a SECOND dlerror() call, which returns NULL)."""
    L = _ffi.lib()
    assert L.rtmi_rccl_probe(b"librccl_does_not_exist.so.9") == -2
    msg = L.rtmi_last_error().decode()
    assert "dlopen(librccl_does_not_exist.so.9)" in msg and "cannot open shared object file" in msg
    assert L.rtmi_rccl_probe(b"libm.so.6") == -2 and "lacks ncclCommInitAll" in L.rtmi_last_error().decode()  # opens, but is not an RCCL
    if os.path.exists("/opt/rocm/lib/librccl.so.1"):
        assert L.rtmi_rccl_probe(None) == 0, L.rtmi_last_error()


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.RtmiError) as e:
        r.Context(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "raytrace_clj_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src.replace("no oracle", ""), fn


def _decode_png(data):
    """minimal reader for what save_png writes (8-bit RGB, filter 0)"""
    import struct
    import zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + body) & 0xFFFFFFFF)
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def test_png_and_ppm_writers_round_trip(tmp_path):
    """the reference saves a PNG through imagez (core.clj:76,112); here: stdlib writers"""
    from raytrace_clj_amd import core
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    core.save_png(tmp_path / "a.png", img)
    assert np.array_equal(_decode_png((tmp_path / "a.png").read_bytes()), img)
    core.save_ppm(tmp_path / "a.ppm", img)
    data = (tmp_path / "a.ppm").read_bytes()
    assert data.startswith(b"P6\n53 37\n255\n") and np.array_equal(np.frombuffer(data[len(b"P6\n53 37\n255\n"):], np.uint8).reshape(37, 53, 3), img)


def test_image_map_reads_png(tmp_path):
    """(image-map :filename "earth.png") -- texture.clj:135-138: the PNG decoder against the writer of this package and against a
    hand-filtered file (every scanline filter type, RGBA and palette colour types)"""
    import struct
    import zlib
    from raytrace_clj_amd import core, texture
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (11, 7, 3), dtype=np.uint8)
    core.save_png(tmp_path / "t.png", img)
    assert np.array_equal(texture.image_map(filename=str(tmp_path / "t.png")).image, img)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    def encode(px, ctype, filters, palette=None):  # px [h, w, ch]; apply the given filter type per row
        h, w, ch = px.shape
        rows, prev = [], np.zeros(w * ch, np.int32)
        for y in range(h):
            cur = px[y].reshape(-1).astype(np.int32)
            ft = filters[y % len(filters)]
            a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
            c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
            if ft == 0: pred = 0
            elif ft == 1: pred = a
            elif ft == 2: pred = prev
            elif ft == 3: pred = (a + prev) >> 1
            else:
                p0 = a + prev - c
                pa, pb, pc = abs(p0 - a), abs(p0 - prev), abs(p0 - c)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            rows.append(bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
            prev = cur
        body = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
        if palette is not None:
            body += chunk(b"PLTE", palette.tobytes())
        return body + chunk(b"IDAT", zlib.compress(b"".join(rows))) + chunk(b"IEND", b"")

    rgba = rng.integers(0, 256, (9, 6, 4), dtype=np.uint8)
    assert np.array_equal(texture.decode_png(encode(rgba, 6, [0, 1, 2, 3, 4])), rgba[:, :, :3])
    pal = rng.integers(0, 256, (5, 3), dtype=np.uint8)
    idx = rng.integers(0, 5, (8, 10, 1), dtype=np.uint8)
    assert np.array_equal(texture.decode_png(encode(idx, 3, [4, 3, 1], palette=pal)), pal[idx[:, :, 0]])
    grey = rng.integers(0, 256, (4, 5, 1), dtype=np.uint8)
    assert np.array_equal(texture.decode_png(encode(grey, 0, [2, 4])), np.repeat(grey, 3, axis=2))
    with pytest.raises(ValueError):
        texture.decode_png(b"not a png")


# ---- bench.py's roofline arithmetic (no GPU: synthetic counters) ----------------------------------------------------------
def test_bench_roofline_object_is_consistent():
    """every fraction bench.py prints is a fraction (<= 1) for counter values of the size the GPU reports, the PMC import is
    stamped with the kernel sources it was taken with, and a workload without a profile still yields the schema's keys"""
    import bench
    counts = {"segments": 1.43e9, "samples": 5.3e8, "pixels": 1920 * 1080, "aabb_tests": 4.34e10, "prim_tests": 4.35e9}
    rf = bench.roofline("C3", "bvh", "f64", 10001, 0.0971, 1, counts)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm", "on_chip_fetch", "launch_ms", "launches_per_step", "kernel"):
        assert k in rf, k
    assert rf["bound"] == "valu" and 0 < rf["hbm"]["frac"] < 0.05 and 0 < rf["on_chip_fetch"]["frac"] < 1 and rf["hbm"]["target_40pct_met"] is False
    assert rf["on_chip_fetch"]["as_fraction_of_hbm_peak"] > rf["on_chip_fetch"]["frac"]
    if rf["frac"] is not None:  # profiles/pmc_counters.json holds C3/bvh/f64
        assert 0 < rf["frac"] <= 1 and rf["traffic"] > 0 and rf["counts"]["kernel_sha"] and isinstance(rf["counts"]["stale"], bool)
        assert 0.3 < rf["lanes_active"] <= 1
    none = bench.roofline("C1", "bvh", "f64", 488, 1e-3, 1, dict(counts, segments=2e5, samples=8e4, pixels=2e4, aabb_tests=4e6, prim_tests=6e5))
    assert none["frac"] is None and none["traffic"] is None and none["hbm"]["frac"] < 1
    flat = bench.roofline("C2", "flat", "f64", 488, 9.3e-3, 1, dict(counts, segments=5.24e7, samples=2.05e7, pixels=3.2e5))
    assert "flat_scan_flops" in flat and flat["on_chip_fetch"]["frac"] < 1


def test_half_outward_is_directed_rounding_to_half():
    """the tree's half-plane node records round every box plane OUTWARD to IEEE half (rtmi.hip: half_outward, integer arithmetic on the float's bits): against
    numpy's round-to-nearest half stepped to the neighbour on the required side -- one million floats over the whole half range and beyond, subnormals, signed
    zeros, ties, the overflow threshold, infinities.  Host arithmetic only."""
    L = _ffi.lib()
    rng = np.random.default_rng(3)
    x = np.concatenate([
        (rng.normal(0, 1, 400000) * 10.0 ** rng.integers(-9, 6, 400000)),
        rng.uniform(-70000, 70000, 200000), rng.uniform(-1e-4, 1e-4, 200000), rng.uniform(-2.0 ** -14, 2.0 ** -14, 100000),
        np.arange(-2048, 2049) * 2.0 ** -24, np.arange(-2048, 2049) * 2.0 ** -25, np.float32(65504.0) + np.arange(-64, 65) * np.float32(0.5),
        [0.0, -0.0, 65504.0, 65519.99, 65520.0, 65535.9, 65536.0, 1e30, -1e30, np.inf, -np.inf, 5.9604645e-08, 2.98e-08, 1e-45, -1e-45, 6.1035156e-05, -6.1035156e-05],
    ]).astype(np.float32)
    with np.errstate(over="ignore"):
        h = x.astype(np.float16)  # round to nearest even
        back = h.astype(np.float32)
        up = np.where(back < x, np.nextafter(h, np.float16(np.inf)), h)
        dn = np.where(back > x, np.nextafter(h, np.float16(-np.inf)), h)
    # the signed zero the library returns when nothing is stepped is numpy's too (RN keeps the sign); a stepped zero has a sign by construction
    got_up = np.array([L.rtmi_test_half_outward(float(v), 1) for v in x[:200000]], np.uint16)
    got_dn = np.array([L.rtmi_test_half_outward(float(v), 0) for v in x[:200000]], np.uint16)
    idx = np.concatenate([np.arange(200000), np.arange(len(x) - 9000, len(x))])
    got_up = np.concatenate([got_up, [L.rtmi_test_half_outward(float(v), 1) for v in x[-9000:]]]).astype(np.uint16)
    got_dn = np.concatenate([got_dn, [L.rtmi_test_half_outward(float(v), 0) for v in x[-9000:]]]).astype(np.uint16)
    gu, gd = got_up.view(np.float16), got_dn.view(np.float16)
    xs = x[idx]
    assert np.all(gu.astype(np.float32) >= xs) and np.all(gd.astype(np.float32) <= xs)               # conservative
    assert np.array_equal(gu, up[idx]) and np.array_equal(gd, dn[idx])                                    # ... and the closest such half
    nz = xs != 0
    assert np.array_equal(got_up[nz], up[idx][nz].view(np.uint16)) and np.array_equal(got_dn[nz], dn[idx][nz].view(np.uint16))  # bit for bit away from zero


def test_tree_build_is_the_same_on_one_thread_and_on_the_team():
    """rtmi_scene_create builds the device's tree and the entry grid's rectangle trees (C3: 11 025 of them) on a team of threads: every job builds in a builder
    of its own and the results are appended in job order, so the node array and the grid's root codes do not depend on the thread count.  Host code only."""
    L = _ffi.lib()
    import raytrace_clj_amd as r
    from raytrace_clj_amd import flatten as fl
    for n, cells in ((11, 12), (24, 26)):
        f = fl.flatten(r.scene.make_random_scene(800, 400, n, False))
        geom = np.ascontiguousarray(f.prim_geom[:, :4])
        cam = np.ascontiguousarray(f.cam)
        out = []
        for threads in (1, 8, 1, 8):
            h, info, ms = np.zeros(1, np.uint64), np.zeros(4, np.int32), np.zeros(1)
            assert L.rtmi_test_build_tree(len(geom), _ffi.ptr(geom), _ffi.ptr(cam), threads, _ffi.ptr(h), _ffi.ptr(info), _ffi.ptr(ms)) == 0
            out.append((int(h[0]), tuple(int(v) for v in info)))
        assert len(set(out)) == 1, out
        assert out[0][1][2] == cells and out[0][1][3] == 2 and out[0][1][0] > 4 * cells * cells  # an entry grid of that many cells per side, dome + ground kept out of the tree

"""Host mirror of raytrace-clj.core (src/raytrace_clj/core.clj): the driver.

`render` replaces the render loop of -main (core.clj:100-108): instead of cp/upmap over 32-pixel
chunks calling `pixel` (core.clj:43-57) -> `color` (core.clj:17-41) on the JVM, the scene is flattened
once (flatten.py) and the whole loop runs in the HIP kernels behind librtmi.so.  `main` keeps the
reference's positional CLI `name nx ny ns` (core.clj:73-80) and the progress/summary line
(display.clj:20-24); the cover scene (core.clj:89) is the default scene.

hit / scatter / emitted / sample / get_ray are the protocol entry points (hitable.clj:7-10,
shader.clj:22-24, texture.clj:8-9, camera.clj:5-6): each runs the SAME device function the render
kernel uses, for one call, through the probe entry points of the C-ABI.  Nothing here computes on the CPU."""
import ctypes as C
import sys
import time

import numpy as np

from . import _ffi
from . import camera as cam
from . import flatten as fl
from . import hitable as hitm
from . import shader as shad
from . import texture as texm
from ._ffi import F32, F64, RtmiError, check, ptr

T_MIN = 0.001
T_MAX = 3.4028234663852886e38  # Float/MAX_VALUE, core.clj:25
DEFAULT_DEPTH = 50              # core.clj:20,45
RENDER_SEED = 0x5EED0002

_PRECISION = {"f64": F64, "f32": F32, F64: F64, F32: F32}


class Context:
    """One rtmi_ctx: a HIP device binding (one per process/GPU)."""

    def __init__(self, device=0, timing=False):
        L = _ffi.lib()
        h = C.c_void_p()
        check(L.rtmi_init(int(device), _ffi.FLAG_TIMING if timing else 0, C.byref(h)))
        self.handle = h
        self.device = int(device)

    def set_option(self, name, value):
        check(_ffi.lib().rtmi_set_option(self.handle, name.encode(), int(value)))

    def device_info(self):
        cus, lds, hbm = C.c_int32(), C.c_int32(), C.c_int64()
        arch = C.create_string_buffer(64)
        check(_ffi.lib().rtmi_device_info(self.handle, C.byref(cus), C.byref(lds), C.byref(hbm), arch, 64))
        return {"compute_units": cus.value, "lds_bytes_per_cu": lds.value, "hbm_bytes": hbm.value, "arch": arch.value.decode()}

    def last_trace_ms(self):
        ms, n = C.c_double(), C.c_int32()
        check(_ffi.lib().rtmi_last_trace_ms(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_accel(self):
        """'bvh' or 'flat': what the last render ran (a small mixed-kind scene is scanned even when the tree was asked for: option flat_below)"""
        v = C.c_int32()
        check(_ffi.lib().rtmi_last_accel(self.handle, C.byref(v)))
        return "bvh" if v.value == _ffi.ACCEL_BVH else "flat"

    def last_reduce_ms(self):
        """(sum of reduce_kernel ms, launches) of the window the last last_trace_ms() call closed"""
        ms, n = C.c_double(), C.c_int32()
        check(_ffi.lib().rtmi_last_reduce_ms(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_traversal_counters(self):
        """(AABB slab tests, exact primitive tests) of the last render; needs set_option("count_traversal", 1) before it
        (metrics.clj:10 aabb.intersection.total, for the device's own tree)"""
        a, b = C.c_uint64(), C.c_uint64()
        check(_ffi.lib().rtmi_last_traversal_counters(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self.handle:
            _ffi.lib().rtmi_shutdown(self.handle)
            self.handle = None


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class DeviceScene:
    """A flattened scene resident in HBM (rtmi_scene)."""

    def __init__(self, scene, camera=None, ctx=None):
        self.ctx = ctx or default_context()
        self.flat = scene if isinstance(scene, fl.FlatScene) else fl.flatten(scene, camera)
        f = self.flat
        keep = [np.ascontiguousarray(a, dt) for a, dt in (
            (f.prim_kind, np.int32), (f.prim_geom, np.float64), (f.prim_mat, np.int32),
            (f.mat_kind, np.int32), (f.mat_tex, np.int32), (f.mat_param, np.float64),
            (f.tex_kind, np.int32), (f.tex_param, np.float64), (f.tex_child, np.int32), (f.cam, np.float64))]
        h = C.c_void_p()
        n = len(keep[0])
        flip = np.ascontiguousarray(getattr(f, "prim_flip", np.zeros(n)), np.int32)
        xform = np.ascontiguousarray(getattr(f, "prim_xform", np.zeros((n, 2))), np.int32)
        xk = np.ascontiguousarray(getattr(f, "xform_kind", np.zeros(0)), np.int32)
        xp = np.ascontiguousarray(getattr(f, "xform_param", np.zeros((0, 3))), np.float64)
        if len(flip) != n or len(xform) != n:  # a FlatScene assembled by hand without the instancing arrays
            flip, xform = np.zeros(n, np.int32), np.zeros((n, 2), np.int32)
        check(_ffi.lib().rtmi_scene_create_ex(
            self.ctx.handle, n, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]),
            len(keep[3]), ptr(keep[3]), ptr(keep[4]), ptr(keep[5]),
            len(keep[6]), ptr(keep[6]), ptr(keep[7]), ptr(keep[8]), int(f.cam_kind), ptr(keep[9]),
            ptr(flip), ptr(xform), len(xk), ptr(xk), ptr(xp), C.byref(h)))
        self.handle = h
        if getattr(f, "perlin_vectors", None) is not None:  # perlin.clj:6-17 tables (seeded)
            vec = np.ascontiguousarray(f.perlin_vectors, np.float64)
            perm = np.ascontiguousarray(f.perlin_perm, np.int32)
            check(_ffi.lib().rtmi_scene_set_perlin(h, ptr(vec), ptr(perm)))
        calls = getattr(f, "media_calls", None)
        if calls is not None and len(calls) and getattr(f, "media_mode", 0) == 2:  # Hitlists holding media below bvh-nodes: the call sequence with its narrowing ranges
            calls = np.ascontiguousarray(calls, np.int32)
            lo = np.ascontiguousarray(f.media_narrow_from, np.int32)
            check(_ffi.lib().rtmi_scene_set_media_calls_narrowed(h, len(calls), ptr(calls), ptr(lo)))
        elif calls is not None and len(calls):  # ConstantMedium hit? invocation order (a medium in a one-item bvh leaf is asked twice)
            calls = np.ascontiguousarray(calls, np.int32)
            check(_ffi.lib().rtmi_scene_set_media_calls(h, len(calls), ptr(calls)))
        if getattr(f, "media_mode", 0) == 1:  # the world is a Hitlist holding media: their hit? sees the t-max narrowed by the items before them
            check(_ffi.lib().rtmi_scene_set_media_mode(h, int(f.media_mode)))
        images = getattr(f, "images", None) or []
        if images:  # ImageMap pixels (texture.clj:126-133)
            imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
            wh = np.ascontiguousarray([[im.shape[1], im.shape[0]] for im in imgs], np.int32)
            rgb = np.ascontiguousarray(np.concatenate([im.reshape(-1) for im in imgs]), np.uint8)
            check(_ffi.lib().rtmi_scene_set_images(h, len(imgs), ptr(wh), ptr(rgb)))

    def clone(self, ctx):
        """the same scene replicated onto another context / device (rtmi_scene_clone)"""
        h = C.c_void_p()
        check(_ffi.lib().rtmi_scene_clone(self.handle, ctx.handle, C.byref(h)))
        other = object.__new__(DeviceScene)
        other.ctx, other.flat, other.handle = ctx, self.flat, h
        return other

    def close(self):
        if self.handle:
            _ffi.lib().rtmi_scene_destroy(self.handle)
            self.handle = None

    # ---- the hot path, host buffers (what the JNA host calls) ---------------------------------------
    def render(self, nx, ny, ns, depth=DEFAULT_DEPTH, seed=RENDER_SEED, precision="f64", region=None):
        """-> (linear float64 [h,w,3] = per-pixel mean before sqrt, rgb8 uint8 [h,w,3], counters {total-rays,total-pixels})"""
        x0, y0, x1, y1 = region if region is not None else (0, 0, nx, ny)
        lin = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0), 3), np.float64)
        q = np.zeros(lin.shape, np.uint8)
        cnt = np.zeros(2, np.uint64)
        check(_ffi.lib().rtmi_render(self.handle, nx, ny, ns, depth, seed, _PRECISION[precision], x0, y0, x1, y1,
                                     ptr(lin), ptr(q), ptr(cnt)))
        return lin, q, cnt

    # ---- the hot path, HBM-resident buffers (torch tensors or raw device pointers) --------------------
    def render_device(self, nx, ny, ns, out_linear=None, out_rgb8=None, out_counters=None, depth=DEFAULT_DEPTH,
                      seed=RENDER_SEED, precision="f64", stream=None):
        check(_ffi.lib().rtmi_render_device(self.handle, nx, ny, ns, depth, seed, _PRECISION[precision],
                                            ptr(out_linear), ptr(out_rgb8), ptr(out_counters), ptr(stream)))

    def render_tiles_device(self, nx, ny, ns, tile_first, tile_stride, out_tiles, out_counters=None, depth=DEFAULT_DEPTH,
                            seed=RENDER_SEED, precision="f64", stream=None):
        check(_ffi.lib().rtmi_render_tiles_device(self.handle, nx, ny, ns, depth, seed, _PRECISION[precision],
                                                  tile_first, tile_stride, ptr(out_tiles), ptr(out_counters), ptr(stream)))

    # ---- probes ------------------------------------------------------------------------------------------
    def probe_hit(self, rays, t_min=T_MIN, t_max=T_MAX, precision="f64"):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 7)
        out = np.zeros((len(rays), 11), np.float64)
        check(_ffi.lib().rtmi_probe_hit(self.handle, _PRECISION[precision], len(rays), ptr(rays), t_min, t_max, ptr(out)))
        return out

    def probe_paths(self, rays, keys, depth=DEFAULT_DEPTH, ctr0=0, max_seg=0, precision="f64"):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 7)
        keys = np.ascontiguousarray(keys, np.uint64)
        n = len(rays)
        rgb, nseg, nlog = np.zeros((n, 3)), np.zeros(n, np.uint64), np.zeros(n, np.int32)
        log = np.zeros((n, max_seg, _ffi.SEG_REC)) if max_seg else None
        check(_ffi.lib().rtmi_probe_paths(self.handle, _PRECISION[precision], n, ptr(rays), ptr(keys), ctr0, depth,
                                          ptr(rgb), ptr(nseg), ptr(log), max_seg, ptr(nlog)))
        return rgb, nseg, log, nlog

    def probe_camera(self, uv, keys, precision="f64"):
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        keys = np.ascontiguousarray(keys, np.uint64)
        out = np.zeros((len(uv), 8))
        check(_ffi.lib().rtmi_probe_camera(self.handle, _PRECISION[precision], len(uv), ptr(uv), ptr(keys), ptr(out)))
        return out

    def probe_texture(self, tex, uvp, precision="f64"):
        uvp = np.ascontiguousarray(uvp, np.float64).reshape(-1, 5)
        out = np.zeros((len(uvp), 3))
        check(_ffi.lib().rtmi_probe_texture(self.handle, _PRECISION[precision], int(tex), len(uvp), ptr(uvp), ptr(out)))
        return out

    def probe_scatter(self, mat, rays, hits, keys, precision="f64"):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 7)
        hits = np.ascontiguousarray(hits, np.float64).reshape(-1, 8)
        keys = np.ascontiguousarray(keys, np.uint64)
        out = np.zeros((len(rays), 9))
        check(_ffi.lib().rtmi_probe_scatter(self.handle, _PRECISION[precision], int(mat), len(rays), ptr(rays), ptr(hits),
                                            ptr(keys), ptr(out)))
        return out


def probe_rng(key, d0, n, precision="f64", ctx=None):
    ctx = ctx or default_context()
    bits, real = np.zeros(n, np.uint64), np.zeros(n, np.float64)
    check(_ffi.lib().rtmi_probe_rng(ctx.handle, _PRECISION[precision], key, d0, n, ptr(bits), ptr(real)))
    return bits, real


def probe_arith(abc, ctx=None):
    ctx = ctx or default_context()
    abc = np.ascontiguousarray(abc, np.float64).reshape(-1, 3)
    out = np.zeros_like(abc)
    check(_ffi.lib().rtmi_probe_arith(ctx.handle, len(abc), ptr(abc), ptr(out)))
    return out


def probe_math(abc, tmin=0.001, tmax=3.4028234663852886e38, ctx=None):
    """[n, 12]: sqrt(a), atan2(a, b), asin(a), sphere u, v of (a, b, c), a / b by the per-ray reciprocal, a / (2 pi), reciprocal path taken,
    the traversal's float upper bound of c, the medium's log(a), a / b by the refined reciprocal of a signed divisor, that path taken"""
    ctx = ctx or default_context()
    abc = np.ascontiguousarray(abc, np.float64).reshape(-1, 3)
    slots = 12
    out = np.zeros((len(abc), slots))
    check(_ffi.lib().rtmi_probe_math2(ctx.handle, len(abc), ptr(abc), tmin, tmax, slots, ptr(out)))
    return out


def sample_key(seed, pixel, sample):
    return int(_ffi.lib().rtmi_sample_key(seed, pixel, sample))


# ---- protocol entry points ------------------------------------------------------------------------------
_dummy_camera = None


def _one_off(world_items, camera=None):
    global _dummy_camera
    if camera is None:
        if _dummy_camera is None:
            _dummy_camera = cam.PinholeCamera(np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3))
        camera = _dummy_camera
    return DeviceScene(hitm.Hitlist(list(world_items)), camera)


def _ray7(r):
    return np.concatenate([np.asarray(r["origin"], np.float64), np.asarray(r["direction"], np.float64), [float(r["time"])]])


def hit(obj, r, t_min, t_max):
    """(hit? obj r t-min t-max) -> {:t :p :uv :normal :material} or None (hitable.clj:7-10), evaluated on the device."""
    items = [obj] if not isinstance(obj, (list, tuple)) else obj
    ds = _one_off(items)
    try:
        o = ds.probe_hit(_ray7(r), float(t_min), float(t_max))[0]
        leaves = []
        fl._leaves(items, leaves, set())
    finally:
        ds.close()
    if o[0] == 0:
        return None
    return {"t": o[2], "p": o[3:6].copy(), "uv": (o[9], o[10]), "normal": o[6:9].copy(), "material": leaves[int(o[1])][0].material}


class _Holder(hitm.Sphere):
    pass


def scatter(material, ray_in, hrec, key=0):
    """(scatter material ray-in hrec) -> {:scattered ray :attenuation vec3} or None (shader.clj:22-24); the draws the
    reference takes from (rand) come from stream `key`."""
    ds = _one_off([hitm.Sphere(np.zeros(3), 1.0, material)])
    try:
        uv = hrec.get("uv", (0.0, 0.0))
        h = np.concatenate([np.asarray(hrec["p"], np.float64), np.asarray(hrec["normal"], np.float64), [uv[0], uv[1]]])
        o = ds.probe_scatter(0, _ray7(ray_in), h, np.array([key], np.uint64))[0]
    finally:
        ds.close()
    if o[0] == 0:
        return None
    return {"scattered": {"origin": np.asarray(hrec["p"], np.float64), "direction": o[1:4].copy(), "time": o[7]},
            "attenuation": o[4:7].copy()}


def emitted(material, uv, p):
    """(emitted material uv p) (shader.clj:22-24): zero except DiffuseLight = (sample tex uv p)."""
    if isinstance(material, shad.DiffuseLight):
        return sample(material.tex, uv, p)
    return np.zeros(3)


def sample(tex, uv, p):
    """(sample tex uv p) (texture.clj:8-9), evaluated on the device."""
    ds = _one_off([hitm.Sphere(np.zeros(3), 1.0, shad.Lambertian(tex))])
    try:
        t = int(ds.flat.mat_tex[0])
        return ds.probe_texture(t, np.concatenate([[uv[0], uv[1]], np.asarray(p, np.float64)]))[0]
    finally:
        ds.close()


def get_ray(camera, u, v, key=0):
    """(get-ray camera u v) (camera.clj:5-6), evaluated on the device; lens/time draws come from stream `key`."""
    ds = _one_off([], camera)
    try:
        o = ds.probe_camera(np.array([u, v], np.float64), np.array([key], np.uint64))[0]
    finally:
        ds.close()
    return {"origin": o[0:3].copy(), "direction": o[3:6].copy(), "time": o[6]}


# ---- driver --------------------------------------------------------------------------------------------
def render(scene, nx, ny, ns, depth=DEFAULT_DEPTH, seed=RENDER_SEED, precision="f64", ctx=None):
    """Render {:camera :world} -> (linear, rgb8, counters).  Replaces core.clj:100-108."""
    ds = DeviceScene(scene, ctx=ctx)
    try:
        return ds.render(nx, ny, ns, depth, seed, precision)
    finally:
        ds.close()


def save_ppm(path, rgb8):
    """imagez `save` (core.clj:112) has no PPM writer; binary P6 written here."""
    h, w, _ = rgb8.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgb8, np.uint8).tobytes())


def save_png(path, rgb8):
    """imagez `save` (core.clj:112) writes whatever the extension says, PNG by default (core.clj:76): 8-bit RGB, no alpha,
    one zlib stream, filter 0 on every row (stdlib only)."""
    import struct
    import zlib
    a = np.ascontiguousarray(rgb8, np.uint8)
    h, w, _ = a.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


SCENES = {  # the scene choices of core.clj:82-90 (there: commented-out lines; here: the 5th argument)
    "random": lambda s, nx, ny: s.make_random_scene(nx, ny, 11, True),      # core.clj:89, the Shirley cover scene
    "final": lambda s, nx, ny: s.make_final(nx, ny),                        # core.clj:90, the line that is active as shipped
    "two-spheres": lambda s, nx, ny: s.make_two_spheres(nx, ny),            # core.clj:82
    "two-perlin-spheres": lambda s, nx, ny: s.make_two_perlin_spheres(nx, ny),
    "textured-sphere": lambda s, nx, ny: s.make_textured_sphere(nx, ny),
    "subsurface-sphere": lambda s, nx, ny: s.make_subsurface_sphere(nx, ny),
    "two-triangles": lambda s, nx, ny: s.make_two_triangles(nx, ny),
    "example-light": lambda s, nx, ny: s.make_example_light(nx, ny),
    "cornell-box": lambda s, nx, ny: s.make_cornell_box(nx, ny, False),     # core.clj:88 passes classic = false
    "cornell-box-classic": lambda s, nx, ny: s.make_cornell_box(nx, ny, True),
}


def main(argv=None):
    """lein-run compatible: `name nx ny ns [win|scene]` (core.clj:73-80).  The reference picks its scene by editing the
    source (core.clj:82-90); here the 5th argument names it (default: the cover scene; "true"/"win", the reference's
    window switch, is accepted and ignored -- there is no display on this path)."""
    from . import scene as scenes
    argv = list(sys.argv[1:] if argv is None else argv)
    name = argv[0] if len(argv) > 0 else "render.png"  # core.clj:76
    nx = int(argv[1]) if len(argv) > 1 else 200
    ny = int(argv[2]) if len(argv) > 2 else 100
    nr = int(argv[3]) if len(argv) > 3 else 100
    which = argv[4] if len(argv) > 4 and argv[4] not in ("true", "win") else "random"
    if which not in SCENES:
        raise SystemExit("unknown scene %r; one of %s" % (which, ", ".join(sorted(SCENES))))
    tstart = time.time()
    sc = SCENES[which](scenes, nx, ny)
    lin, rgb8, cnt = render(sc, nx, ny, nr)
    elapsed = time.time() - tstart
    print("%.2fs, %d%%, ETA %.2fs" % (elapsed, 100, 0.0))  # display.clj:20-24
    print("total-rays %d total-pixels %d" % (int(cnt[0]), int(cnt[1])))  # metrics.clj:8-9
    if name.lower().endswith(".ppm"):
        save_ppm(name, rgb8)
    elif name.lower().endswith(".npy"):
        np.save(name, rgb8)
    else:
        save_png(name, rgb8)
    print("wrote", name)  # core.clj:113
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""ctypes loader for the CPU oracle (oracle/rt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (raytrace_clj_amd) never imports this module.

The scene argument of every function is duck-typed: any object with the flat-array attributes
prim_kind/prim_geom/prim_mat, mat_kind/mat_tex/mat_param, tex_kind/tex_param/tex_child,
cam_kind/cam (numpy arrays, the layout include/rtmi.h documents).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SEG_REC = 12


def build(force=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    so = os.path.join(_HERE, "librt_oracle.so")
    src = os.path.join(_HERE, "rt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return so


class _Scene(C.Structure):
    _fields_ = [
        ("n_prims", C.c_int32), ("prim_kind", C.c_void_p), ("prim_geom", C.c_void_p), ("prim_mat", C.c_void_p),
        ("n_mats", C.c_int32), ("mat_kind", C.c_void_p), ("mat_tex", C.c_void_p), ("mat_param", C.c_void_p),
        ("n_tex", C.c_int32), ("tex_kind", C.c_void_p), ("tex_param", C.c_void_p), ("tex_child", C.c_void_p),
        ("cam_kind", C.c_int32), ("cam", C.c_void_p),
        ("n_nodes", C.c_int32), ("node_kind", C.c_void_p), ("node_a", C.c_void_p), ("node_d", C.c_void_p),
        ("node_prim", C.c_void_p), ("node_children", C.c_void_p), ("root", C.c_int32),
        ("perlin_vec", C.c_void_p), ("perlin_perm", C.c_void_p), ("n_images", C.c_int32), ("image_wh", C.c_void_p),
        ("image_off", C.c_void_p), ("image_rgb", C.c_void_p),
    ]


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One loaded oracle library (precision 'f64' or 'f32')."""

    def __init__(self, precision="f64"):
        build()
        name = "librt_oracle.so" if precision == "f64" else "librt_oracle_f32.so"
        self.lib = C.CDLL(os.path.join(_HERE, name))
        L = self.lib
        L.rto_sample_key.restype = C.c_uint64
        L.rto_sample_key.argtypes = [C.c_uint64] * 3
        L.rto_draw_bits.restype = C.c_uint64
        L.rto_draw_bits.argtypes = [C.c_uint64] * 2
        L.rto_draw.restype = C.c_double
        L.rto_draw.argtypes = [C.c_uint64] * 2
        L.rto_schlick.restype = C.c_double
        L.rto_schlick.argtypes = [C.c_double] * 2
        L.rto_render.restype = C.c_int
        L.rto_render.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_uint64] + [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_int]
        L.rto_refract.restype = C.c_int
        L.rto_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.rto_aabb_hit.restype = C.c_int
        L.rto_aabb_hit.argtypes = [C.c_void_p] * 4 + [C.c_double] * 2
        self.precision = precision

    # -- scene marshalling ------------------------------------------------------------------
    def _scene(self, fs):
        keep = dict(
            prim_kind=_i32(fs.prim_kind), prim_geom=_f64(fs.prim_geom), prim_mat=_i32(fs.prim_mat),
            mat_kind=_i32(fs.mat_kind), mat_tex=_i32(fs.mat_tex), mat_param=_f64(fs.mat_param),
            tex_kind=_i32(fs.tex_kind), tex_param=_f64(fs.tex_param), tex_child=_i32(fs.tex_child),
            cam=_f64(fs.cam),
        )
        s = _Scene()
        s.n_prims = len(keep["prim_kind"])
        s.n_mats = len(keep["mat_kind"])
        s.n_tex = len(keep["tex_kind"])
        s.cam_kind = int(fs.cam_kind)
        for k, v in keep.items():
            setattr(s, k, v.ctypes.data)
        tree = getattr(fs, "tree", None)  # optional nested world (oracle/tree.py)
        if tree is not None:
            tk = dict(node_kind=_i32(tree["kind"]), node_a=_i32(tree["a"]), node_d=_f64(tree["d"]), node_prim=_i32(tree["prim"]),
                      node_children=_i32(tree["children"]))
            s.n_nodes = len(tk["node_kind"])
            s.root = int(tree["root"])
            for k, v in tk.items():
                setattr(s, k, v.ctypes.data)
            keep.update(tk)
        if getattr(fs, "perlin_vectors", None) is not None:
            keep["perlin_vec"] = _f64(fs.perlin_vectors)
            keep["perlin_perm"] = _i32(fs.perlin_perm)
            s.perlin_vec, s.perlin_perm = keep["perlin_vec"].ctypes.data, keep["perlin_perm"].ctypes.data
        images = getattr(fs, "images", None) or []
        if images:
            imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
            keep["image_wh"] = _i32([[im.shape[1], im.shape[0]] for im in imgs])
            sizes = [im.size for im in imgs]
            keep["image_off"] = np.ascontiguousarray(np.concatenate([[0], np.cumsum(sizes)[:-1]]), np.int64)
            keep["image_rgb"] = np.ascontiguousarray(np.concatenate([im.reshape(-1) for im in imgs]), np.uint8)
            s.n_images = len(imgs)
            s.image_wh, s.image_off, s.image_rgb = (keep[k].ctypes.data for k in ("image_wh", "image_off", "image_rgb"))
        s._keep = keep
        return s

    # -- the path ---------------------------------------------------------------------------
    def render(self, fs, nx, ny, ns, depth=50, seed=0, region=None, nthreads=1):
        """Returns (linear float64 [h,w,3], rgb8 uint8 [h,w,3], counters uint64 [2]); row 0 = top."""
        x0, y0, x1, y1 = region if region is not None else (0, 0, nx, ny)
        lin = np.zeros((y1 - y0, x1 - x0, 3), np.float64)
        q = np.zeros((y1 - y0, x1 - x0, 3), np.uint8)
        cnt = np.zeros(2, np.uint64)
        s = self._scene(fs)
        rc = self.lib.rto_render(C.byref(s), nx, ny, ns, depth, seed, x0, y0, x1, y1, _p(lin), _p(q), _p(cnt), nthreads)
        if rc != 0:
            raise ValueError("rto_render rc=%d" % rc)
        return lin, q, cnt

    def probe_hit(self, fs, rays, tmin=0.001, tmax=3.4028234663852886e38, bvh_seed=None):
        rays = _f64(rays).reshape(-1, 7)
        out = np.zeros((len(rays), 11), np.float64)
        s = self._scene(fs)
        if bvh_seed is None:
            self.lib.rto_probe_hit(C.byref(s), len(rays), _p(rays), C.c_double(tmin), C.c_double(tmax), _p(out))
        else:
            self.lib.rto_probe_hit_bvh(C.byref(s), C.c_uint64(bvh_seed), len(rays), _p(rays), C.c_double(tmin),
                                       C.c_double(tmax), _p(out))
        return out

    def probe_paths(self, fs, rays, keys, depth=50, ctr0=0, max_seg=0):
        rays = _f64(rays).reshape(-1, 7)
        keys = _u64(keys)
        n = len(rays)
        rgb = np.zeros((n, 3), np.float64)
        nseg = np.zeros(n, np.uint64)
        log = np.zeros((n, max_seg, SEG_REC), np.float64) if max_seg else None
        nlog = np.zeros(n, np.int32)
        s = self._scene(fs)
        self.lib.rto_probe_paths(C.byref(s), n, _p(rays), _p(keys), C.c_uint64(ctr0), depth, _p(rgb), _p(nseg),
                                 _p(log) if max_seg else None, max_seg, _p(nlog))
        return rgb, nseg, log, nlog

    def probe_camera(self, fs, uv, keys):
        uv = _f64(uv).reshape(-1, 2)
        keys = _u64(keys)
        out = np.zeros((len(uv), 8), np.float64)
        s = self._scene(fs)
        self.lib.rto_probe_camera(C.byref(s), len(uv), _p(uv), _p(keys), _p(out))
        return out

    def probe_texture(self, fs, tex, uvp):
        uvp = _f64(uvp).reshape(-1, 5)
        out = np.zeros((len(uvp), 3), np.float64)
        s = self._scene(fs)
        self.lib.rto_probe_texture(C.byref(s), int(tex), len(uvp), _p(uvp), _p(out))
        return out

    def probe_scatter(self, fs, mat, rays, hits, keys):
        rays = _f64(rays).reshape(-1, 7)
        hits = _f64(hits).reshape(-1, 8)
        keys = _u64(keys)
        out = np.zeros((len(rays), 9), np.float64)
        s = self._scene(fs)
        self.lib.rto_probe_scatter(C.byref(s), int(mat), len(rays), _p(rays), _p(hits), _p(keys), _p(out))
        return out

    # -- small pure functions ------------------------------------------------------------------
    def sample_key(self, seed, pix, s):
        return int(self.lib.rto_sample_key(seed, pix, s))

    def draw_bits(self, key, d):
        return int(self.lib.rto_draw_bits(key, d))

    def draw(self, key, d):
        return float(self.lib.rto_draw(key, d))

    def schlick(self, cosine, ri):
        return float(self.lib.rto_schlick(cosine, ri))

    def point_at_parameter(self, o, d, t):
        o, d, out = _f64(o), _f64(d), np.zeros(3)
        self.lib.rto_point_at_parameter(_p(o), _p(d), C.c_double(t), _p(out))
        return out

    def reflect(self, v, n):
        v, n, out = _f64(v), _f64(n), np.zeros(3)
        self.lib.rto_reflect(_p(v), _p(n), _p(out))
        return out

    def refract(self, v, n, ni_over_nt):
        v, n, out = _f64(v), _f64(n), np.zeros(3)
        ok = self.lib.rto_refract(_p(v), _p(n), C.c_double(ni_over_nt), _p(out))
        return out if ok else None

    def center_at_time(self, c0, t0, c1, t1, t):
        c0, c1, out = _f64(c0), _f64(c1), np.zeros(3)
        self.lib.rto_center_at_time(_p(c0), C.c_double(t0), _p(c1), C.c_double(t1), C.c_double(t), _p(out))
        return out

    def sphere_uv(self, n):
        n, out = _f64(n), np.zeros(2)
        self.lib.rto_sphere_uv(_p(n), _p(out))
        return out

    def quantise(self, mean):
        mean, out = _f64(mean), np.zeros(3, np.uint8)
        self.lib.rto_quantise(_p(mean), _p(out))
        return out

    def make_camera(self, kind, lookfrom, lookat, vup, vfov, aspect, aperture=0.0, focus_dist=1.0, t0=0.0, t1=0.0):
        a, b, c, cam = _f64(lookfrom), _f64(lookat), _f64(vup), np.zeros(24)
        self.lib.rto_make_camera(int(kind), _p(a), _p(b), _p(c), C.c_double(vfov), C.c_double(aspect),
                                 C.c_double(aperture), C.c_double(focus_dist), C.c_double(t0), C.c_double(t1), _p(cam))
        return cam

    def aabb_hit(self, vmin, vmax, o, d, tmin, tmax):
        vmin, vmax, o, d = _f64(vmin), _f64(vmax), _f64(o), _f64(d)
        return bool(self.lib.rto_aabb_hit(_p(vmin), _p(vmax), _p(o), _p(d), tmin, tmax))

    def prim_bbox(self, fs, i, t0=0.0, t1=0.0):
        vmin, vmax = np.zeros(3), np.zeros(3)
        s = self._scene(fs)
        self.lib.rto_prim_bbox(C.byref(s), int(i), C.c_double(t0), C.c_double(t1), _p(vmin), _p(vmax))
        return vmin, vmax

    def surrounding_bbox(self, b0, b1):
        a0, a1, c0, c1 = _f64(b0[0]), _f64(b0[1]), _f64(b1[0]), _f64(b1[1])
        vmin, vmax = np.zeros(3), np.zeros(3)
        self.lib.rto_surrounding_bbox(_p(a0), _p(a1), _p(c0), _p(c1), _p(vmin), _p(vmax))
        return vmin, vmax

"""Host mirror of raytrace-clj.hitable (src/raytrace_clj/hitable.clj), the records in scope of the
GPU path: Hitlist (15-26), AABB (36-53), bvh-node / make-bvh (97-123), UVSphere (141-177),
Sphere (180-216), MovingSphere (224-264).

`bbox` and `make-bvh` are scene construction and stay on the host; `hit?` runs on the device
(`hit(obj, r, t_min, t_max)` below is the protocol entry point and routes to the device probe).
ConstantMedium (516-543) draws a random number inside hit?, so its draw order is the order the reference visits the
world's leaves: the depth-first order of bvh-node / wrappers, which is what the flattener preserves.  It is supported
when no Hitlist encloses it (a Hitlist narrows t-max between siblings, which changes what a medium draws), as in every
scene of scene.clj.  A world containing any other record is rejected by the flattener with UnsupportedOnGpuPath."""
import math
from dataclasses import dataclass
from typing import Any, List

import numpy as np

from .util import SplitMix64


class Hitable:
    """(defprotocol Hitable (hit? [this r t-min t-max]) (bbox [this t0 t1])) -- hitable.clj:7-10"""

    def hit(self, r, t_min, t_max):
        from . import core
        return core.hit(self, r, t_min, t_max)


@dataclass(eq=False)
class Hitlist(Hitable):  # hitable.clj:15 -- no bbox method, like the reference
    items: List[Any]


@dataclass(eq=False)
class AABB:  # hitable.clj:36
    vmin: np.ndarray
    vmax: np.ndarray


@dataclass(eq=False)
class bvh_node(Hitable):  # hitable.clj:97
    left: Any
    right: Any
    box: AABB

    def bbox(self, t0, t1):
        return self.box


@dataclass(eq=False)
class UVSphere(Hitable):  # hitable.clj:141
    center: np.ndarray
    radius: float
    material: Any

    def bbox(self, t0, t1):  # hitable.clj:169-172
        r = np.full(3, self.radius)
        return AABB(self.center - r, self.center + r)


@dataclass(eq=False)
class Sphere(Hitable):  # hitable.clj:180
    center: np.ndarray
    radius: float
    material: Any

    def bbox(self, t0, t1):  # hitable.clj:208-211
        r = np.full(3, self.radius)
        return AABB(self.center - r, self.center + r)


def center_at_time(center0, t0, center1, t1, t):
    """(mat/lerp center0 center1 (/ (- t t0) (- t1 t0))) -- hitable.clj:219-222"""
    f = (t - t0) / (t1 - t0)
    return np.asarray(center0, np.float64) * (1.0 - f) + np.asarray(center1, np.float64) * f


@dataclass(eq=False)
class MovingSphere(Hitable):  # hitable.clj:224
    center0: np.ndarray
    t0: float
    center1: np.ndarray
    t1: float
    radius: float
    material: Any

    def bbox(self, t_start, t_end):  # hitable.clj:253-259
        r = np.full(3, self.radius)
        cs = center_at_time(self.center0, self.t0, self.center1, self.t1, t_start)
        ce = center_at_time(self.center0, self.t0, self.center1, self.t1, t_end)
        return make_surrounding_bbox(AABB(cs - r, cs + r), AABB(ce - r, ce + r))


@dataclass(eq=False)
class RectXY(Hitable):  # hitable.clj:269
    x0: float
    y0: float
    x1: float
    y1: float
    k: float
    material: Any

    def bbox(self, t_start, t_end):  # hitable.clj:292-294
        return AABB(np.array([self.x0, self.y0, self.k - 0.0001]), np.array([self.x1, self.y1, self.k + 0.0001]))


@dataclass(eq=False)
class RectXZ(Hitable):  # hitable.clj:301
    x0: float
    z0: float
    x1: float
    z1: float
    k: float
    material: Any

    def bbox(self, t_start, t_end):  # hitable.clj:324-326
        return AABB(np.array([self.x0, self.k - 0.0001, self.z0]), np.array([self.x1, self.k + 0.0001, self.z1]))


@dataclass(eq=False)
class RectYZ(Hitable):  # hitable.clj:333
    y0: float
    z0: float
    y1: float
    z1: float
    k: float
    material: Any

    def bbox(self, t_start, t_end):  # hitable.clj:356-358
        return AABB(np.array([self.k - 0.0001, self.y0, self.z0]), np.array([self.k + 0.0001, self.y1, self.z1]))


@dataclass(eq=False)
class FlipNormals(Hitable):  # hitable.clj:375
    item: Any

    def bbox(self, t_start, t_end):
        return self.item.bbox(t_start, t_end)


@dataclass(eq=False)
class Translate(Hitable):  # hitable.clj:391
    item: Any
    offset: np.ndarray

    def bbox(self, t_start, t_end):  # hitable.clj:397-400
        b = self.item.bbox(t_start, t_end)
        return AABB(b.vmin + self.offset, b.vmax + self.offset)


@dataclass(eq=False)
class RotateY(Hitable):  # hitable.clj:410
    obj: Any
    rotated_bbox: AABB
    sin_theta: float
    cos_theta: float

    def bbox(self, t_start, t_end):
        return self.rotated_bbox


@dataclass(eq=False)
class Box(Hitable):  # hitable.clj:491
    p0: np.ndarray
    p1: np.ndarray
    sides: Hitlist

    def bbox(self, t_start, t_end):
        return AABB(self.p0, self.p1)


@dataclass(eq=False)
class Triangle(Hitable):  # hitable.clj:548
    v0: np.ndarray
    v1: np.ndarray
    v2: np.ndarray
    material: Any

    def bbox(self, t_start, t_end):  # hitable.clj:572-577
        eps = np.full(3, 0.0001)
        return AABB(np.minimum(np.minimum(self.v0, self.v1), self.v2) - eps, np.maximum(np.maximum(self.v0, self.v1), self.v2) + eps)


@dataclass(eq=False)
class ConstantMedium(Hitable):  # hitable.clj:516
    boundary: Any
    density: float
    phase_fn: Any

    def bbox(self, t_start, t_end):
        return self.boundary.bbox(t_start, t_end)


def constant_medium(*, boundary, density, albedo):
    """(constant-medium :boundary b :density d :albedo tex) -- hitable.clj:543-546"""
    from .shader import isotropic
    return ConstantMedium(boundary, float(density), isotropic(albedo=albedo))


def rect_xy(*, x0, y0, x1, y1, k, material):
    """(rect-xy :x0 :y0 :x1 :y1 :k :material) -- hitable.clj:296-299"""
    return RectXY(float(x0), float(y0), float(x1), float(y1), float(k), material)


def rect_xz(*, x0, z0, x1, z1, k, material):
    """(rect-xz ...) -- hitable.clj:328-331"""
    return RectXZ(float(x0), float(z0), float(x1), float(z1), float(k), material)


def rect_yz(*, y0, z0, y1, z1, k, material):
    """(rect-yz ...) -- hitable.clj:360-363"""
    return RectYZ(float(y0), float(z0), float(y1), float(z1), float(k), material)


def flip_normals(*, item):
    """(flip-normals :item x) -- hitable.clj:383-386"""
    return FlipNormals(item)


def translate(*, item, offset):
    """(translate :item x :offset v) -- hitable.clj:402-405"""
    return Translate(item, np.asarray(offset, np.float64))


def make_rotate_y(obj, theta):
    """hitable.clj:452-481: sin/cos of theta degrees and the box of the 8 rotated corners of (bbox obj 0 1)"""
    radians = float(theta) * (math.pi / 180.0)
    cos_th, sin_th = math.cos(radians), math.sin(radians)
    b = obj.bbox(0, 1)
    fmax = 3.4028234663852886e38
    new_min, new_max = np.full(3, fmax), np.full(3, -fmax)
    for x in (b.vmin[0], b.vmax[0]):
        for y in (b.vmin[1], b.vmax[1]):
            for z in (b.vmin[2], b.vmax[2]):
                c = np.array([cos_th * x + sin_th * z, y, (-(sin_th * x)) + cos_th * z])
                new_min, new_max = np.minimum(new_min, c), np.maximum(new_max, c)
    return RotateY(obj, AABB(new_min, new_max), sin_th, cos_th)


def rotate_y(*, item, theta):
    """(rotate-y :item x :theta degrees) -- hitable.clj:483-486"""
    return make_rotate_y(item, theta)


def box(*, p0, p1, material):
    """(box :p0 :p1 :material) -- hitable.clj:496-511: six rectangles, three of them with flipped normals"""
    p0, p1 = np.asarray(p0, np.float64), np.asarray(p1, np.float64)
    x0, y0, z0 = (float(v) for v in p0)
    x1, y1, z1 = (float(v) for v in p1)
    return Box(p0, p1, Hitlist([
        RectXY(x0, y0, x1, y1, z1, material),
        FlipNormals(RectXY(x0, y0, x1, y1, z0, material)),
        RectXZ(x0, z0, x1, z1, y1, material),
        FlipNormals(RectXZ(x0, z0, x1, z1, y0, material)),
        RectYZ(y0, z0, y1, z1, x1, material),
        FlipNormals(RectYZ(y0, z0, y1, z1, x0, material)),
    ]))


def triangle(*, v0, v1, v2, material):
    """(triangle :v0 :v1 :v2 :material) -- hitable.clj:579-581"""
    return Triangle(np.asarray(v0, np.float64), np.asarray(v1, np.float64), np.asarray(v2, np.float64), material)


def make_surrounding_bbox(box0, box1):
    """hitable.clj:87-92"""
    return AABB(np.minimum(box0.vmin, box1.vmin), np.maximum(box0.vmax, box1.vmax))


def hitlist(*, items):
    """(hitlist :items xs) -- hitable.clj:28-31"""
    return Hitlist(list(items))


def aabb(*, vmin, vmax):
    """(aabb :vmin :vmax) -- hitable.clj:50-53"""
    return AABB(np.asarray(vmin, np.float64), np.asarray(vmax, np.float64))


def uv_sphere(*, center, radius, material):
    """(uv-sphere :center :radius :material) -- hitable.clj:174-177"""
    return UVSphere(np.asarray(center, np.float64), float(radius), material)


def sphere(*, center, radius, material):
    """(sphere :center :radius :material) -- hitable.clj:213-216"""
    return Sphere(np.asarray(center, np.float64), float(radius), material)


def moving_sphere(*, center0, t0, center1, t1, radius, material):
    """(moving-sphere :center0 :t0 :center1 :t1 :radius :material) -- hitable.clj:261-264"""
    return MovingSphere(np.asarray(center0, np.float64), float(t0), np.asarray(center1, np.float64), float(t1),
                        float(radius), material)


def make_bvh(hitable_list, t0, t1, rng=None):
    """(make-bvh hitable-list t0 t1) -- hitable.clj:108-123.  The reference picks the split axis with
    the unseeded (rand-int 3); here it comes from `rng` (a SplitMix64) so scenes are reproducible."""
    rng = rng if rng is not None else SplitMix64(0x5EED0003)
    axis = rng.rand_int(3)
    my_list = sorted(hitable_list, key=lambda h: h.bbox(t0, t1).vmin[axis])
    n = len(my_list)
    if n == 1:
        L = my_list[0]
        return bvh_node(L, L, L.bbox(t0, t1))
    if n == 2:
        L, R = my_list
        return bvh_node(L, R, make_surrounding_bbox(L.bbox(t0, t1), R.bbox(t0, t1)))
    h = (n + 1) // 2  # (split-at (/ n 2) ...): a Ratio for odd n, take/drop count past it
    L = make_bvh(my_list[:h], t0, t1, rng)
    R = make_bvh(my_list[h:], t0, t1, rng)
    return bvh_node(L, R, make_surrounding_bbox(L.bbox(t0, t1), R.bbox(t0, t1)))

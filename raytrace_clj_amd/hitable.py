"""Host mirror of raytrace-clj.hitable (src/raytrace_clj/hitable.clj), the records in scope of the
GPU path: Hitlist (15-26), AABB (36-53), bvh-node / make-bvh (97-123), UVSphere (141-177),
Sphere (180-216), MovingSphere (224-264).

`bbox` and `make-bvh` are scene construction and stay on the host; `hit?` runs on the device
(`hit(obj, r, t_min, t_max)` below is the protocol entry point and routes to the device probe).
Records outside this list (rectangles, boxes, instances, media, triangles) are not mirrored: a world
containing anything else is rejected by the flattener with UnsupportedOnGpuPath."""
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

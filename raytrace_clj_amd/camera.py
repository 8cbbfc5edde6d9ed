"""Host mirror of raytrace-clj.camera (src/raytrace_clj/camera.clj).

The constructors (camera.clj:18-33, 50-66) stay on the host, in the reference's operation order
(core.matrix ops restated with numpy float64: normalise = v * (1/|v|), cross, scalar*vector);
`get-ray` (camera.clj:8-16, 35-48) runs on the device."""
import math
from dataclasses import dataclass

import numpy as np


class Camera:
    """(defprotocol Camera (get-ray [this u v])) -- camera.clj:5-6"""

    def get_ray(self, u, v, key=0):
        from . import core
        return core.get_ray(self, u, v, key)


@dataclass(eq=False)
class PinholeCamera(Camera):  # camera.clj:8
    origin: np.ndarray
    lleft: np.ndarray
    horiz: np.ndarray
    vert: np.ndarray


@dataclass(eq=False)
class ThinLensCamera(Camera):  # camera.clj:35
    origin: np.ndarray
    lleft: np.ndarray
    horiz: np.ndarray
    vert: np.ndarray
    u: np.ndarray
    v: np.ndarray
    w: np.ndarray
    aperture: float
    t0: float
    t1: float


def _normalise(a):
    d = math.sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2])
    return a * (1.0 / d) if d > 0 else a.copy()


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], np.float64)


def _basis(lookfrom, lookat, vup, vfov, aspect):
    theta = float(vfov) * (math.pi / 180.0)
    half_height = math.tan(theta / 2.0)
    half_width = float(aspect) * half_height
    w = _normalise(lookfrom - lookat)
    u = _normalise(_cross(vup, w))
    v = _cross(w, u)
    return half_height, half_width, u, v, w


def pinhole_camera(*, lookfrom, lookat, vup, vfov, aspect):
    """(pinhole-camera :lookfrom :lookat :vup :vfov :aspect) -- camera.clj:18-33"""
    lookfrom, lookat, vup = (np.asarray(x, np.float64) for x in (lookfrom, lookat, vup))
    hh, hw, u, v, w = _basis(lookfrom, lookat, vup, vfov, aspect)
    return PinholeCamera(lookfrom, lookfrom - ((u * hw + v * hh) + w), u * (2.0 * hw), v * (2.0 * hh))


def thin_lens_camera(*, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist, t0, t1):
    """(thin-lens-camera ...) -- camera.clj:50-66"""
    lookfrom, lookat, vup = (np.asarray(x, np.float64) for x in (lookfrom, lookat, vup))
    hh, hw, u, v, w = _basis(lookfrom, lookat, vup, vfov, aspect)
    fd = float(focus_dist)
    lleft = lookfrom - ((u * (fd * hw) + v * (fd * hh)) + w * fd)
    return ThinLensCamera(lookfrom, lleft, u * ((2.0 * fd) * hw), v * ((2.0 * fd) * hh), u, v, w,
                          float(aperture), float(t0), float(t1))

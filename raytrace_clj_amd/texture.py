"""Host mirror of raytrace-clj.texture (src/raytrace_clj/texture.clj): the Texture records a
scene is built from.  `sample` (texture.clj:8-9) is evaluated on the device; `sample(tex, uv, p)`
here is the protocol entry point and routes to the device probe."""
from dataclasses import dataclass

import numpy as np


class Texture:
    """(defprotocol Texture (sample [this uv p])) -- texture.clj:8-9"""

    def sample(self, uv, p):
        from . import core
        return core.sample(self, uv, p)


@dataclass(eq=False)
class Constant(Texture):  # texture.clj:14-16
    color: np.ndarray


@dataclass(eq=False)
class UVGradient(Texture):  # texture.clj:26-34
    co: np.ndarray
    cu: np.ndarray
    cv: np.ndarray
    cuv: np.ndarray


@dataclass(eq=False)
class Checkerboard(Texture):  # texture.clj:44-50
    tex0: Texture
    tex1: Texture
    scale: float


def constant(*, color):
    """(constant :color c) -- texture.clj:18-21"""
    return Constant(np.asarray(color, np.float64))


def uv_gradient(*, co, cu, cv, cuv):
    """(uv-gradient :co :cu :cv :cuv) -- texture.clj:36-39"""
    return UVGradient(*(np.asarray(x, np.float64) for x in (co, cu, cv, cuv)))


def checkerboard(*, tex0, tex1, scale):
    """(checkerboard :tex0 :tex1 :scale) -- texture.clj:52-55"""
    return Checkerboard(tex0, tex1, float(scale))

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


@dataclass(eq=False)
class PerlinNoise(Texture):  # texture.clj:60-64
    scale: float


@dataclass(eq=False)
class PerlinTurbulence(Texture):  # texture.clj:74-78
    scale: float
    depth: int


@dataclass(eq=False)
class Marble(Texture):  # texture.clj:88-93
    scale: float
    depth: int


@dataclass(eq=False)
class FlipTextureU(Texture):  # texture.clj:103-106
    tex: Texture


@dataclass(eq=False)
class FlipTextureV(Texture):  # texture.clj:113-116
    tex: Texture


@dataclass(eq=False)
class ImageMap(Texture):  # texture.clj:126-133; `image` is an [h, w, 3] uint8 array (row 0 = top) instead of a BufferedImage
    image: np.ndarray


def perlin_noise(*, scale):
    """(perlin-noise :scale s) -- texture.clj:66-69"""
    return PerlinNoise(float(scale))


def perlin_turbulence(*, scale, depth):
    """(perlin-turbulence :scale s :depth d) -- texture.clj:80-83"""
    return PerlinTurbulence(float(scale), int(depth))


def marble(*, scale, depth):
    """(marble :scale s :depth d) -- texture.clj:95-98"""
    return Marble(float(scale), int(depth))


def flip_texture_u(*, tex):
    """(flip-texture-u :tex t) -- texture.clj:108-111"""
    return FlipTextureU(tex)


def flip_texture_v(*, tex):
    """(flip-texture-v :tex t) -- texture.clj:118-121"""
    return FlipTextureV(tex)


def image_map(*, filename=None, image=None):
    """(image-map :filename f) -- texture.clj:135-138.  The reference loads the file with imagez; this mirror takes the
    decoded pixels (`image`, [h, w, 3] uint8) or a binary PPM (P6) file name -- there is no PNG decoder on this path."""
    if image is None:
        with open(filename, "rb") as fh:
            data = fh.read()
        parts = data.split(None, 4)
        if parts[0] != b"P6":
            raise ValueError("image_map reads binary PPM (P6) files; decode other formats to an array and pass image=")
        w, h = int(parts[1]), int(parts[2])
        image = np.frombuffer(parts[4][: w * h * 3], np.uint8).reshape(h, w, 3)
    return ImageMap(np.ascontiguousarray(image, np.uint8))


def constant(*, color):
    """(constant :color c) -- texture.clj:18-21"""
    return Constant(np.asarray(color, np.float64))


def uv_gradient(*, co, cu, cv, cuv):
    """(uv-gradient :co :cu :cv :cuv) -- texture.clj:36-39"""
    return UVGradient(*(np.asarray(x, np.float64) for x in (co, cu, cv, cuv)))


def checkerboard(*, tex0, tex1, scale):
    """(checkerboard :tex0 :tex1 :scale) -- texture.clj:52-55"""
    return Checkerboard(tex0, tex1, float(scale))

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


def decode_png(data):
    """8-bit, non-interlaced PNG (greyscale, RGB, palette, with or without alpha) -> [h, w, 3] uint8, row 0 = top.  What imagez'
    load-image (texture.clj:137) hands get-pixel; stdlib only (zlib + the five scanline filters)."""
    import struct
    import zlib
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, palette, w, h, ctype = 8, [], None, 0, 0, -1
    while pos + 8 <= len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 4, 6):
                raise ValueError("PNG: only 8-bit non-interlaced images are decoded (depth %d, colour type %d, interlace %d)" % (depth, ctype, interlace))
        elif tag == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):  # undo the scanline filters (PNG spec 9.2); bpp = ch bytes
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(w * ch, np.int32)
            for x in range(w * ch):
                a = cur[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                elif ft == 4:
                    p0 = a + b - c
                    pa, pb, pc = abs(p0 - a), abs(p0 - b), abs(p0 - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError("PNG: unknown filter type %d" % ft)
                cur[x] = (line[x] + pred) & 255
        out[y] = cur
        prev = cur
    px = out.reshape(h, w, ch)
    if ctype == 3:
        if palette is None:
            raise ValueError("PNG: palette image without PLTE")
        return np.ascontiguousarray(palette[px[:, :, 0]])
    if ctype in (0, 4):
        return np.ascontiguousarray(np.repeat(px[:, :, :1], 3, axis=2))
    return np.ascontiguousarray(px[:, :, :3])


def image_map(*, filename=None, image=None):
    """(image-map :filename f) -- texture.clj:135-138.  The reference loads the file with imagez; this mirror takes the
    decoded pixels (`image`, [h, w, 3] uint8), a PNG (decode_png) or a binary PPM (P6) file name."""
    if image is None:
        with open(filename, "rb") as fh:
            data = fh.read()
        if data[:8] == b"\x89PNG\r\n\x1a\n":
            image = decode_png(data)
        else:
            parts = data.split(None, 4)
            if parts[0] != b"P6":
                raise ValueError("image_map reads PNG and binary PPM (P6) files; decode other formats to an array and pass image=")
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

"""Host mirror of raytrace-clj.shader (src/raytrace_clj/shader.clj): the Shader records.
`scatter` / `emitted` (shader.clj:22-24) are evaluated on the device; the methods here are the
protocol entry points and route to the device probe."""
from dataclasses import dataclass

from .texture import Texture


class Shader:
    """(defprotocol Shader (scatter [this ray-in hrec]) (emitted [this uv p])) -- shader.clj:22-24"""

    def scatter(self, ray_in, hrec, key=0):
        from . import core
        return core.scatter(self, ray_in, hrec, key)


@dataclass(eq=False)
class Lambertian(Shader):  # shader.clj:29-36
    albedo: Texture


@dataclass(eq=False)
class Metal(Shader):  # shader.clj:46-59
    albedo: Texture
    fuzz: float


@dataclass(eq=False)
class Dielectric(Shader):  # shader.clj:76-104
    ri: float


@dataclass(eq=False)
class DiffuseLight(Shader):  # shader.clj:114-119
    tex: Texture


@dataclass(eq=False)
class Isotropic(Shader):  # shader.clj:129-138 (the phase function of ConstantMedium)
    albedo: Texture


def isotropic(*, albedo):
    """(isotropic :albedo tex) -- shader.clj:140-143"""
    return Isotropic(albedo)


def lambertian(*, albedo):
    """(lambertian :albedo tex) -- shader.clj:38-41"""
    return Lambertian(albedo)


def metal(*, albedo, fuzz):
    """(metal :albedo tex :fuzz f) -- shader.clj:61-64 (fuzz is not clamped)"""
    return Metal(albedo, float(fuzz))


def dielectric(*, ri):
    """(dielectric :ri n) -- shader.clj:106-109"""
    return Dielectric(float(ri))


def diffuse_light(*, tex):
    """(diffuse-light :tex tex) -- shader.clj:121-124"""
    return DiffuseLight(tex)

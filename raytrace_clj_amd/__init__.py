"""raytrace_clj_amd: MI355X-native (gfx950) implementation of raytrace-clj's per-pixel Monte-Carlo
sampling path behind the reference's Hitable / Shader / Texture / Camera protocol surface.

Python host mirror of the reference namespaces (util, hitable, shader, texture, camera, scene, core) over the
C-ABI in include/rtmi.h (raytrace_clj_amd/csrc -> raytrace_clj_amd/lib/librtmi.so).  Compute happens only in
the HIP kernels; there is no CPU fallback."""
from . import camera, core, flatten, hitable, scene, shader, texture, util  # noqa: F401
from .core import Context, DeviceScene, render  # noqa: F401
from .flatten import FlatScene, UnsupportedOnGpuPath  # noqa: F401

__all__ = ["camera", "core", "flatten", "hitable", "scene", "shader", "texture", "util", "Context", "DeviceScene",
           "render", "FlatScene", "UnsupportedOnGpuPath"]

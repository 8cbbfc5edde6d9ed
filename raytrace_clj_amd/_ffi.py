"""ctypes binding of librtmi.so (include/rtmi.h).  There is no CPU fallback: if the HIP library is
missing or no gfx950 device is visible, every entry point raises."""
import ctypes as C
import os

import numpy as np

# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7, + its own HSA runtime).
# Two HIP runtimes in one process cannot both own the GPU ("No HIP GPUs are available"), so when torch is the
# process's device-memory/stream/collective plumbing it must be loaded FIRST: librtmi.so's NEEDED libamdhip64.so.7
# then binds to the runtime that is already mapped.  (A torch-free host, e.g. the JVM via JNA, gets /opt/rocm's.)
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover - torch-free host
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTMI_LIB") or os.path.join(_HERE, "lib", "librtmi.so")  # RTMI_LIB: experiment builds

# every symbol include/rtmi.h declares
SYMBOLS = [
    "rtmi_last_error", "rtmi_backend_name", "rtmi_version", "rtmi_init", "rtmi_shutdown", "rtmi_set_option",
    "rtmi_device_info", "rtmi_scene_create", "rtmi_scene_create_ex", "rtmi_scene_set_perlin", "rtmi_scene_set_images", "rtmi_scene_set_media_calls", "rtmi_scene_set_media_mode", "rtmi_scene_set_media_calls_narrowed", "rtmi_scene_device_bytes", "rtmi_scene_destroy", "rtmi_render", "rtmi_render_device",
    "rtmi_render_tiles_device", "rtmi_local_tiles", "rtmi_assemble_device", "rtmi_last_trace_ms", "rtmi_last_reduce_ms", "rtmi_probe_hit",
    "rtmi_probe_paths", "rtmi_probe_camera", "rtmi_probe_texture", "rtmi_probe_scatter", "rtmi_probe_rng",
    "rtmi_sample_key", "rtmi_test_half_outward", "rtmi_test_build_tree", "rtmi_probe_arith", "rtmi_probe_math", "rtmi_probe_math2", "rtmi_last_traversal_counters",
    "rtmi_scene_clone", "rtmi_render_multi", "rtmi_render_multi_device", "rtmi_last_gather_ms",
    "rtmi_last_gather_path", "rtmi_rccl_probe", "rtmi_stream_idle", "rtmi_last_passes", "rtmi_last_accel",
]

F64, F32 = 0, 1
ACCEL_FLAT, ACCEL_BVH = 0, 1
FLAG_TIMING = 1
GATHER_PATHS = {0: "none", 1: "same-device", 2: "peer-copy", 3: "rccl"}
SEG_REC = 12
TILE = 8


class RtmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rtmi error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load librtmi.so (built in-tree by `make -C raytrace_clj_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtmiError(-2, "HIP extension %s is missing; build it with `make -C raytrace_clj_amd/csrc` "
                            "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    L.rtmi_last_error.restype = C.c_char_p
    L.rtmi_backend_name.restype = C.c_char_p
    L.rtmi_version.restype = C.c_int
    L.rtmi_init.argtypes = [C.c_int, C.c_uint32, C.POINTER(vp)]
    L.rtmi_shutdown.argtypes = [vp]
    L.rtmi_set_option.argtypes = [vp, C.c_char_p, i64]
    L.rtmi_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.c_char_p, i32]
    L.rtmi_scene_create.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, C.POINTER(vp)]
    L.rtmi_scene_create_ex.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, C.POINTER(vp)]
    L.rtmi_scene_set_perlin.argtypes = [vp, vp, vp]
    L.rtmi_scene_set_images.argtypes = [vp, i32, vp, vp]
    L.rtmi_scene_set_media_calls.argtypes = [vp, i32, vp]
    L.rtmi_scene_set_media_mode.argtypes = [vp, i32]
    L.rtmi_scene_set_media_calls_narrowed.argtypes = [vp, i32, vp, vp]
    L.rtmi_scene_device_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
    L.rtmi_scene_destroy.argtypes = [vp]
    L.rtmi_render.argtypes = [vp, i32, i32, i32, i32, u64, i32, i32, i32, i32, i32, vp, vp, vp]
    L.rtmi_render_device.argtypes = [vp, i32, i32, i32, i32, u64, i32, vp, vp, vp, vp]
    L.rtmi_render_tiles_device.argtypes = [vp, i32, i32, i32, i32, u64, i32, i32, i32, vp, vp, vp]
    L.rtmi_local_tiles.argtypes = [i32, i32, i32, i32]
    L.rtmi_local_tiles.restype = i32
    L.rtmi_assemble_device.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp]
    L.rtmi_last_trace_ms.argtypes = [vp, C.POINTER(dbl), C.POINTER(i32)]
    L.rtmi_last_reduce_ms.argtypes = [vp, C.POINTER(dbl), C.POINTER(i32)]
    L.rtmi_probe_hit.argtypes = [vp, i32, i32, vp, dbl, dbl, vp]
    L.rtmi_probe_paths.argtypes = [vp, i32, i32, vp, vp, u64, i32, vp, vp, vp, i32, vp]
    L.rtmi_probe_camera.argtypes = [vp, i32, i32, vp, vp, vp]
    L.rtmi_probe_texture.argtypes = [vp, i32, i32, i32, vp, vp]
    L.rtmi_probe_scatter.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    L.rtmi_probe_rng.argtypes = [vp, i32, u64, u64, i32, vp, vp]
    L.rtmi_sample_key.argtypes = [u64, u64, u64]
    L.rtmi_sample_key.restype = u64
    L.rtmi_probe_arith.argtypes = [vp, i32, vp, vp]
    L.rtmi_test_build_tree.argtypes = [i32, vp, vp, i32, vp, vp, vp]
    L.rtmi_test_half_outward.argtypes = [C.c_double, i32]
    L.rtmi_probe_math.argtypes = [vp, i32, vp, C.c_double, C.c_double, vp]
    L.rtmi_probe_math2.argtypes = [vp, i32, vp, C.c_double, C.c_double, i32, vp]
    L.rtmi_last_traversal_counters.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.rtmi_scene_clone.argtypes = [vp, vp, C.POINTER(vp)]
    L.rtmi_render_multi.argtypes = [i32, C.POINTER(vp), i32, i32, i32, i32, u64, i32, vp, vp, vp]
    L.rtmi_render_multi_device.argtypes = [i32, C.POINTER(vp), i32, i32, i32, i32, u64, i32, vp, vp, vp]
    L.rtmi_last_gather_ms.argtypes = [vp, C.POINTER(dbl)]
    L.rtmi_last_gather_path.argtypes = [vp, C.POINTER(i32)]
    L.rtmi_rccl_probe.argtypes = [C.c_char_p]
    L.rtmi_stream_idle.argtypes = [vp, C.POINTER(i32)]
    L.rtmi_last_passes.argtypes = [vp, C.POINTER(i32)]
    L.rtmi_last_accel.argtypes = [vp, C.POINTER(i32)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("rtmi_version",):
            fn.restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RtmiError(rc, lib().rtmi_last_error().decode("utf-8", "replace"))


def ptr(a):
    """device or host pointer of a numpy array / torch tensor / int / None"""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError("cannot take a pointer of %r" % type(a))

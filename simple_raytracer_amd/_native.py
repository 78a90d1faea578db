"""ctypes binding of libsrt.so (include/srt.h).

There is no fallback: if the library has not been built, or no gfx950 device is usable, the
calls below raise.  `load(path)` can bind any library exporting the same ABI (the test-suite
binds its CPU harness of the kernel source that way).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.environ.get("SRT_LIBRARY") or os.path.join(_HERE, "csrc", "libsrt.so")  # SRT_LIBRARY: experiments only

OK, E_INVALID, E_NO_DEVICE, E_HIP, E_STATE, E_REFERENCE, E_LIMIT = 0, -1, -2, -3, -4, -5, -6
WORLD_BVHNODE, WORLD_SURFACELIST = 0, 1
MODE_COLOR, MODE_NORMAL, MODE_DEPTH, MODE_LIGHTS, MODE_DIFFUSE, MODE_REFLECTION = 0, 1, 2, 3, 4, 5
LAYOUT_FRAME, LAYOUT_TILES = 0, 1
MAX_DEPTH = 16


class SrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libsrt error %d: %s" % (code, message))
        self.code = code


class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("metalness", C.c_float), ("smoothness", C.c_float), ("emission", C.c_float * 3)]


class PointLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("brightness", C.c_float), ("color", C.c_float * 3)]


GATHER_RCCL, GATHER_PEER_COPY = 0, 1
MAX_DEVICES, ALL_DEVICES = 8, -1


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_devices", C.c_int32), ("devices", C.c_int32 * MAX_DEVICES), ("gather", C.c_int32),
                ("reserved", C.c_int32 * 5)]


class DeviceInfo(C.Structure):
    _fields_ = [("n_devices", C.c_int32), ("devices", C.c_int32 * MAX_DEVICES), ("gather", C.c_int32), ("rccl_ranks", C.c_int32),
                ("reserved", C.c_int32 * 5)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("max_depth", C.c_int32),
                ("color_clamp", C.c_float), ("seed", C.c_uint32), ("mode", C.c_int32), ("layout", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays_path", C.c_uint64), ("rays_shadow", C.c_uint64), ("hit_nodes", C.c_uint64), ("box_tests", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("rect_tests", C.c_uint64),
                ("kernel_ms", C.c_double), ("build_ms", C.c_double),
                ("bvh_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("n_prims", C.c_uint32), ("walk_tune", C.c_uint32),
                ("trace_sections", C.c_uint64), ("trace_lanes", C.c_uint64), ("light_sections", C.c_uint64), ("light_lanes", C.c_uint64),
                ("section_cycles", C.c_uint64 * 16), ("section_lane_cycles", C.c_uint64 * 16),
                ("shadow_unasked", C.c_uint64), ("gather_ms", C.c_double), ("assemble_ms", C.c_double),
                ("wavefront_items", C.c_uint64), ("wavefront_nodes", C.c_uint64), ("wavefront_given_up", C.c_uint64), ("wavefront_twins", C.c_uint64)]


# every symbol include/srt.h declares: (name, restype, argtypes)
_FP = C.POINTER(C.c_float)
_UP = C.POINTER(C.c_uint32)
_CTX = C.c_void_p
SYMBOLS = [
    ("srt_version", C.c_int, []),
    ("srt_create", C.c_int, [C.POINTER(Config), C.POINTER(_CTX)]),
    ("srt_destroy", None, [_CTX]),
    ("srt_last_error", C.c_char_p, [_CTX]),
    ("srt_get_device_info", C.c_int, [_CTX, C.POINTER(DeviceInfo)]),
    ("srt_set_camera", C.c_int, [_CTX, _FP, _FP, _FP, _FP]),
    ("srt_set_materials", C.c_int, [_CTX, C.POINTER(Material), C.c_uint32]),
    ("srt_set_spheres", C.c_int, [_CTX, _FP, _FP, _FP, _FP, _UP, C.c_uint32]),
    ("srt_set_rects", C.c_int, [_CTX, _FP, _FP, _FP, _FP, _FP, _UP, C.c_uint32]),
    ("srt_set_triangles", C.c_int, [_CTX, _FP, _FP, _UP, C.POINTER(C.c_uint8), C.c_uint32]),
    ("srt_set_lights", C.c_int, [_CTX, C.POINTER(PointLight), C.c_uint32]),
    ("srt_set_surface_order", C.c_int, [_CTX, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.c_uint32]),
    ("srt_commit_scene", C.c_int, [_CTX, C.c_int]),
    ("srt_render", C.c_int, [_CTX, C.POINTER(RenderParams), C.c_void_p]),
    ("srt_render_device", C.c_int, [_CTX, C.POINTER(RenderParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    ("srt_render_device_static", C.c_int, [_CTX, C.POINTER(RenderParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    ("srt_render_counted", C.c_int, [_CTX, C.POINTER(RenderParams), C.c_void_p]),
    ("srt_render_async", C.c_int, [_CTX, C.POINTER(RenderParams), C.c_void_p]),
    ("srt_poll", C.c_int, [_CTX, C.POINTER(C.c_int)]),
    ("srt_wait", C.c_int, [_CTX]),
    ("srt_render_progressive", C.c_int, [_CTX, C.POINTER(RenderParams), C.c_void_p, C.c_int]),
    ("srt_progress", C.c_int, [_CTX, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("srt_read_preview", C.c_int, [_CTX, C.c_void_p]),
    ("srt_render_mix_buffers", C.c_int, [_CTX, C.POINTER(RenderParams), C.c_int32, C.c_void_p]),
    ("srt_blur_array", C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32]),
    ("srt_contrast_blur_array", C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32]),
    ("srt_despeckle", C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32]),
    ("srt_get_stats", C.c_int, [_CTX, C.POINTER(Stats)]),
    ("srt_get_device_stats", C.c_int, [_CTX, C.c_int32, C.POINTER(Stats)]),
    ("srt_local_pixels", C.c_int64, [C.POINTER(RenderParams)]),
    ("srt_assemble_tiles", C.c_int, [_CTX, C.POINTER(RenderParams), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
]

_cache = {}


def load(path=None, strict=True):
    """Binds the library at `path` (default: the in-tree libsrt.so).  With strict=False symbols the library does
    not export are skipped (the test-suite's CPU harness implements only the synchronous calls)."""
    path = os.path.abspath(path or DEFAULT_LIBRARY)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise ImportError("%s is missing: build it with `make -C simple_raytracer_amd/csrc` (hipcc, gfx950). "
                          "There is no CPU fallback." % path)
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so, and a process that loads the
    # system copy first (through this library) and torch's copy later ends up with a torch that sees no GPU.
    # Importing torch first makes both resolve to the same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        if not strict and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _cache[path] = lib
    return lib

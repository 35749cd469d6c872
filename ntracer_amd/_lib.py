"""ctypes binding of libntracer_hip.so (the C ABI in include/ntracer_hip.h).

The library is the product: if it is missing this module raises at import of
the first symbol -- there is no Python or CPU fallback for the ray-cast path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NTRACER_HIP_LIB") or os.path.join(_HERE, "libntracer_hip.so")   # override: A/B builds

NT_OK = 0
NT_ABORTED = 1
NT_E_INVALID = -1
NT_E_BUSY = -2
NT_E_LOCKED = -3
NT_E_DEVICE = -4
NT_E_NOMEM = -5
NT_E_UNSUPPORTED = -6

NT_MAX_DIM = 64
NT_BATCH_SIZE = 4
KIND_BATCH, KIND_TRIANGLE, KIND_SOLID = 0, 1, 2

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


class LockedError(Exception):
    """render.LockedError (reference src/render.cpp:1326-1336)."""


class NtChannel(C.Structure):
    _fields_ = [("f_r", C.c_float), ("f_g", C.c_float), ("f_b", C.c_float), ("f_c", C.c_float),
                ("bit_size", C.c_uint8), ("tfloat", C.c_uint8), ("_pad", C.c_uint8 * 2)]


class NtImageFormat(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("pitch", C.c_int32), ("nchannels", C.c_int32),
                ("channels", C.POINTER(NtChannel)), ("reversed", C.c_int32)]


class NtMaterial(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("specular", C.c_float * 3), ("opacity", C.c_float),
                ("reflectivity", C.c_float), ("specular_intensity", C.c_float), ("specular_exp", C.c_float)]


class NtSceneDesc(C.Structure):
    _fields_ = [("dimension", C.c_int32), ("root", C.c_int32), ("n_nodes", C.c_int32),
                ("node_axis", i32p), ("node_split", f32p), ("node_left", i32p), ("node_right", i32p),
                ("n_items", C.c_int32), ("items", i32p),
                ("n_batches", C.c_int32), ("batch_recs", f32p), ("batch_mats", i32p),
                ("n_triangles", C.c_int32), ("tri_recs", f32p), ("tri_mats", i32p),
                ("n_solids", C.c_int32), ("solid_recs", f32p), ("solid_types", i32p), ("solid_mats", i32p),
                ("n_materials", C.c_int32), ("materials", C.POINTER(NtMaterial)),
                ("aabb_start", f32p), ("aabb_end", f32p)]


class NtSceneParams(C.Structure):
    _fields_ = [("shadows", C.c_int32), ("camera_light", C.c_int32), ("max_reflect_depth", C.c_int32),
                ("bg_gradient_axis", C.c_int32), ("ambient", C.c_float * 3), ("bg1", C.c_float * 3),
                ("bg2", C.c_float * 3), ("bg3", C.c_float * 3),
                ("n_point_lights", C.c_int32), ("point_light_pos", f32p), ("point_light_color", f32p),
                ("n_global_lights", C.c_int32), ("global_light_dir", f32p), ("global_light_color", f32p)]


class NtRenderOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("band_rank", C.c_int32), ("band_world", C.c_int32), ("band_rows", C.c_int32),
                ("compact", C.c_int32), ("strict_reference", C.c_int32), ("collect_stats", C.c_int32),
                ("overlapped", C.c_int32), ("abort_device", C.c_void_p)]


class NtStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("rays", "shadow_rays", "branches", "leaves", "simplex_tests",
                                          "solid_tests", "hits", "aabb_enter")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class NtKdTreeParams(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("split_threshold", C.c_int32), ("traversal_cost", C.c_float),
                ("intersection_cost", C.c_float)]


class NtKdTree(C.Structure):
    _fields_ = [("root", C.c_int32), ("n_nodes", C.c_int32), ("n_leaf_items", C.c_int32),
                ("node_axis", i32p), ("node_split", f32p), ("node_left", i32p), ("node_right", i32p),
                ("leaf_items", i32p), ("aabb", f32p)]


# every symbol include/ntracer_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("nt_version", C.c_char_p, []),
    ("nt_last_error", C.c_char_p, []),
    ("nt_device_count", C.c_int, []),
    ("nt_box_scene_create", C.c_void_p, [C.c_int]),
    ("nt_composite_scene_create", C.c_void_p, [C.POINTER(NtSceneDesc)]),
    ("nt_scene_destroy", None, [C.c_void_p]),
    ("nt_scene_dimension", C.c_int, [C.c_void_p]),
    ("nt_scene_is_composite", C.c_int, [C.c_void_p]),
    ("nt_scene_set_camera", C.c_int, [C.c_void_p, f32p, f32p]),
    ("nt_scene_get_camera", C.c_int, [C.c_void_p, f32p, f32p]),
    ("nt_scene_set_fov", C.c_int, [C.c_void_p, C.c_float]),
    ("nt_scene_get_fov", C.c_float, [C.c_void_p]),
    ("nt_scene_set_params", C.c_int, [C.c_void_p, C.POINTER(NtSceneParams)]),
    ("nt_scene_lock", C.c_int, [C.c_void_p]),
    ("nt_scene_unlock", C.c_int, [C.c_void_p]),
    ("nt_scene_locked", C.c_int, [C.c_void_p]),
    ("nt_format_bytes_per_pixel", C.c_int, [C.POINTER(NtImageFormat)]),
    ("nt_render", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(NtImageFormat), C.POINTER(NtRenderOpts),
                            C.POINTER(C.c_int)]),
    ("nt_render_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(NtImageFormat),
                                   C.POINTER(NtRenderOpts), C.c_void_p]),
    ("nt_render_frames_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, f32p, f32p,
                                          C.POINTER(NtImageFormat), C.POINTER(NtRenderOpts), C.c_void_p]),
    ("nt_camera_table_create", C.c_void_p, [C.c_int, C.c_int, f32p, f32p, C.c_int]),
    ("nt_camera_table_destroy", None, [C.c_void_p]),
    ("nt_camera_table_frames", C.c_int, [C.c_void_p]),
    ("nt_render_table_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int,
                                         C.POINTER(NtImageFormat), C.POINTER(NtRenderOpts), C.c_void_p]),
    ("nt_calculate_color", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
    ("nt_colors_at", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, i32p, i32p, f32p, C.c_int]),
    ("nt_scene_last_stats", C.c_int, [C.c_void_p, C.POINTER(NtStats)]),
    ("nt_kdtree_build", C.c_int, [C.c_int, C.c_int, f32p, f32p, i32p, f32p, C.POINTER(NtKdTreeParams), C.POINTER(NtKdTree)]),
    ("nt_kdtree_free", None, [C.POINTER(NtKdTree)]),
    ("nt_polytope_clip_box", C.c_int, [C.c_int, C.c_int, f32p, C.POINTER(C.c_uint64), C.c_int, C.c_int, f32p, f32p, f32p, f32p]),
]

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.  Two HIP runtimes in
    one process cannot both open the GPU, so when torch is installed we load ITS runtime first; our
    library's NEEDED libamdhip64.so.7 then binds to that copy by soname.  Without torch (a plain C++
    host) the system runtime in /opt/rocm is used.  Set NTRACER_HIP_SYSTEM_RUNTIME=1 to skip this."""
    if os.environ.get("NTRACER_HIP_SYSTEM_RUNTIME") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load the HIP library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libntracer_hip.so is missing (%s): build it with `python -m ntracer_amd.build` "
                "(hipcc --offload-arch=gfx950).  The ray-cast path has no CPU fallback." % LIB_PATH)
        _preload_torch_hip_runtime()
        if os.environ.get("NTRACER_HIP_LIB"):
            # an A/B or ablation build stands in for the product: say so, every time (a stale variable on a bench box would
            # otherwise be measured as if it were the library)
            import sys
            sys.stderr.write("ntracer_amd: NTRACER_HIP_LIB is set -- loading %s instead of the in-tree libntracer_hip.so\n" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)      # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error():
    return lib().nt_last_error().decode("utf-8", "replace")


def check(status):
    """Map nt_status to the exception the reference raises for the same condition
    (PY_EXCEPT_HANDLERS, src/py_common.hpp:39-48)."""
    if status >= 0:
        return status
    msg = last_error()
    if status == NT_E_INVALID:
        raise ValueError(msg)
    if status == NT_E_BUSY:
        raise RuntimeError(msg)
    if status == NT_E_LOCKED:
        raise LockedError(msg)
    if status == NT_E_NOMEM:
        raise MemoryError(msg)
    if status == NT_E_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)

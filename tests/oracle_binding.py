"""ctypes binding of oracle/libntracer_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (the oracle is the checker, never a fallback for the HIP path).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(_HERE, "..", "oracle")
_LIB = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


class Channel(C.Structure):
    _fields_ = [("f_r", C.c_float), ("f_g", C.c_float), ("f_b", C.c_float), ("f_c", C.c_float),
                ("bit_size", C.c_int32), ("tfloat", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [
        ("is_composite", C.c_int32), ("n", C.c_int32), ("origin", f32p), ("axes", f32p), ("fov", C.c_float),
        ("root", C.c_int32), ("n_nodes", C.c_int32), ("node_axis", i32p), ("node_split", f32p),
        ("node_left", i32p), ("node_right", i32p), ("items", i32p), ("batch_size", C.c_int32),
        ("batch_recs", f32p), ("batch_mats", i32p), ("tri_recs", f32p), ("tri_mats", i32p),
        ("solid_recs", f32p), ("solid_types", i32p), ("solid_mats", i32p), ("materials", f32p),
        ("aabb_start", f32p), ("aabb_end", f32p),
        ("shadows", C.c_int32), ("camera_light", C.c_int32), ("max_reflect_depth", C.c_int32),
        ("bg_gradient_axis", C.c_int32),
        ("ambient", C.c_float * 3), ("bg1", C.c_float * 3), ("bg2", C.c_float * 3), ("bg3", C.c_float * 3),
        ("n_point_lights", C.c_int32), ("pl_pos", f32p), ("pl_color", f32p),
        ("n_global_lights", C.c_int32), ("gl_dir", f32p), ("gl_color", f32p), ("clean_normals", C.c_int32), ("prune_beyond_hit", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("rays", "shadow_rays", "branches", "leaves", "batch_tests",
                                          "simplex_tests", "solid_tests", "hits", "aabb_enter")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libntracer_oracle.so")
        src = os.path.join(ORACLE_DIR, "ntracer_oracle.c")
        if os.environ.get("NTRACER_ORACLE_LIB"):        # tools/sanitize.sh: the same source built with -fsanitize=...
            path = os.environ["NTRACER_ORACLE_LIB"]
        elif not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
            build()
        _LIB = C.CDLL(path)
        _LIB.nto_calculate_color.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        _LIB.nto_colors_at.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_int, i32p, i32p, f32p, C.POINTER(Counters)]
        _LIB.nto_pack_pixel.argtypes = [f32p, C.c_int, C.POINTER(Channel), C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        _LIB.nto_render.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(Channel), C.c_int, C.c_int, C.POINTER(Counters)]
        _LIB.nto_render.restype = C.c_int
        _LIB.nto_renderer_create.argtypes = [C.c_int]
        _LIB.nto_renderer_create.restype = C.c_void_p
        _LIB.nto_renderer_destroy.argtypes = [C.c_void_p]
        _LIB.nto_renderer_destroy.restype = None
        _LIB.nto_renderer_threads.argtypes = [C.c_void_p]
        _LIB.nto_renderer_threads.restype = C.c_int
        _LIB.nto_renderer_render.argtypes = [C.c_void_p, C.POINTER(Scene), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(Channel), C.c_int, C.POINTER(Counters)]
        _LIB.nto_renderer_render.restype = C.c_int
        _LIB.nto_renderer_render_frames.argtypes = [C.c_void_p, C.POINTER(Scene), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.POINTER(Channel), C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_double,
                                                    C.POINTER(C.c_double)]
        _LIB.nto_renderer_render_frames.restype = C.c_int
        _LIB.nto_kd_intersects.argtypes = [C.POINTER(Scene), f32p, f32p, C.c_float, C.c_float, C.c_int, C.c_int,
                                           f32p, i32p, i32p, i32p, f32p, f32p, i32p]
        _LIB.nto_kd_intersects.restype = C.c_int
        _LIB.nto_kd_occludes.argtypes = [C.POINTER(Scene), f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_int,
                                         C.c_int, i32p]
        _LIB.nto_kd_occludes.restype = C.c_int
        _LIB.nto_primary_dir.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class OracleScene:
    """Owns numpy copies of every array and exposes the nto_scene struct."""

    def __init__(self, n, origin, axes, fov=0.8, flat=None, params=None, clean_normals=False, prune=False):
        self._keep = {}
        s = Scene()
        s.clean_normals = 1 if clean_normals else 0
        s.prune_beyond_hit = 1 if prune else 0
        s.n = int(n)
        s.fov = float(fov)
        self.n = int(n)
        self._set("origin", _f(origin), s, f32p)
        self._set("axes", _f(axes).reshape(n, n), s, f32p)
        s.is_composite = 0
        s.batch_size = 4
        s.camera_light = 1
        s.max_reflect_depth = 4
        s.bg_gradient_axis = 1
        s.bg1[:] = (1, 1, 1)
        s.bg3[:] = (0, 1, 1)
        if flat is not None:
            s.is_composite = 1
            s.root = int(flat["root"])
            s.n_nodes = len(flat["node_axis"])
            s.batch_size = int(flat.get("batch_size", 4))
            for k, t, ptr in (("node_axis", _i, i32p), ("node_split", _f, f32p), ("node_left", _i, i32p),
                              ("node_right", _i, i32p), ("items", _i, i32p), ("batch_recs", _f, f32p),
                              ("batch_mats", _i, i32p), ("tri_recs", _f, f32p), ("tri_mats", _i, i32p),
                              ("solid_recs", _f, f32p), ("solid_types", _i, i32p), ("solid_mats", _i, i32p),
                              ("materials", _f, f32p), ("aabb_start", _f, f32p), ("aabb_end", _f, f32p)):
                self._set(k, t(flat[k]), s, ptr)
        self.s = s
        if params is not None:
            self.set_params(params)

    def _set(self, name, arr, s, ptr):
        if arr.size == 0:
            arr = np.zeros(1, arr.dtype)
        self._keep[name] = arr
        setattr(s, name, arr.ctypes.data_as(ptr))

    def set_camera(self, origin, axes):
        self._set("origin", _f(origin), self.s, f32p)
        self._set("axes", _f(axes).reshape(self.n, self.n), self.s, f32p)

    def set_params(self, p):
        s = self.s
        n = self.n
        if "fov" in p:
            s.fov = float(p["fov"])
        for k in ("shadows", "camera_light", "max_reflect_depth", "bg_gradient_axis"):
            if k in p:
                setattr(s, k, int(p[k]))
        for k in ("ambient", "bg1", "bg2", "bg3"):
            if k in p:
                getattr(s, k)[:] = [float(v) for v in p[k]]
        if "point_light_pos" in p:
            pos = _f(p["point_light_pos"]).reshape(-1, n)
            s.n_point_lights = len(pos)
            self._set("pl_pos", pos, s, f32p)
            self._set("pl_color", _f(p["point_light_color"]).reshape(-1, 3), s, f32p)
        if "global_light_dir" in p:
            d = _f(p["global_light_dir"]).reshape(-1, n)
            s.n_global_lights = len(d)
            self._set("gl_dir", d, s, f32p)
            self._set("gl_color", _f(p["global_light_color"]).reshape(-1, 3), s, f32p)

    # ---- calls ----
    def colors_at(self, xs, ys, w, h, counters=False):
        xs = _i(xs)
        ys = _i(ys)
        out = np.zeros((len(xs), 3), np.float32)
        c = Counters()
        lib().nto_colors_at(C.byref(self.s), w, h, len(xs), xs.ctypes.data_as(i32p), ys.ctypes.data_as(i32p),
                            out.ctypes.data_as(f32p), C.byref(c))
        return (out, c.as_dict()) if counters else out

    def render(self, w, h, channels, pitch=0, reversed_=False, threads=0, counters=False):
        ch, bpp = make_channels(channels)
        pitch = pitch or w * bpp
        buf = np.zeros((h, pitch), np.uint8)
        c = Counters()
        r = lib().nto_render(C.byref(self.s), buf.ctypes.data, w, h, pitch, len(ch), ch, int(bool(reversed_)),
                             threads, C.byref(c) if counters else None)
        assert r == 0
        return (buf, c.as_dict()) if counters else buf

    def kd_intersects(self, origin, direction, t_near=-3.4028234663852886e38, t_far=3.4028234663852886e38,
                      skip_item=-1, skip_lane=-1):
        o = _f(origin)
        d = _f(direction)
        dist = C.c_float()
        kind = C.c_int32()
        index = C.c_int32()
        lane = C.c_int32()
        no = np.zeros(self.n, np.float32)
        nd = np.zeros(self.n, np.float32)
        nt = C.c_int32()
        r = lib().nto_kd_intersects(C.byref(self.s), o.ctypes.data_as(f32p), d.ctypes.data_as(f32p), t_near, t_far,
                                    skip_item, skip_lane, C.byref(dist), C.byref(kind), C.byref(index),
                                    C.byref(lane), no.ctypes.data_as(f32p), nd.ctypes.data_as(f32p), C.byref(nt))
        if not r:
            return None
        return dict(dist=dist.value, kind=kind.value, index=index.value, lane=lane.value, origin=no, normal=nd,
                    n_transparent=nt.value)

    def kd_occludes(self, origin, direction, distance=3.4028234663852886e38, t_near=-3.4028234663852886e38,
                    t_far=3.4028234663852886e38):
        o = _f(origin)
        d = _f(direction)
        nt = C.c_int32()
        return bool(lib().nto_kd_occludes(C.byref(self.s), o.ctypes.data_as(f32p), d.ctypes.data_as(f32p), distance,
                                          t_near, t_far, -1, -1, C.byref(nt)))

    def primary_dir(self, x, y, w, h):
        out = np.zeros(self.n, np.float32)
        lib().nto_primary_dir(C.byref(self.s), x, y, w, h, out.ctypes.data_as(f32p))
        return out


class OracleRenderer:
    """blocking_renderer of the reference (render.cpp:769-851): worker threads that persist between frames."""

    def __init__(self, threads=-1):
        self._h = lib().nto_renderer_create(int(threads))
        assert self._h
        self.threads = lib().nto_renderer_threads(self._h)      # workers + the caller

    def close(self):
        if self._h:
            lib().nto_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def render(self, scene, w, h, channels, pitch=0, reversed_=False, counters=False):
        ch, bpp = make_channels(channels)
        pitch = pitch or w * bpp
        buf = np.zeros((h, pitch), np.uint8)
        c = Counters()
        r = lib().nto_renderer_render(self._h, C.byref(scene.s), buf.ctypes.data, w, h, pitch, len(ch), ch, int(bool(reversed_)),
                                      C.byref(c) if counters else None)
        assert r == 0
        return (buf, c.as_dict()) if counters else buf

    def render_frames(self, scene, w, h, channels, origins, axes, nframes, max_seconds=0.0):
        """`nframes` frames of the camera sequence inside ONE C call (the last frame stays in the returned buffer);
        returns (buffer, seconds per frame)."""
        ch, bpp = make_channels(channels)
        buf = np.zeros((h, w * bpp), np.uint8)
        o = _f(origins).reshape(-1, scene.n)
        a = _f(axes).reshape(len(o), scene.n, scene.n)
        secs = (C.c_double * nframes)()
        done = lib().nto_renderer_render_frames(self._h, C.byref(scene.s), buf.ctypes.data, w, h, 0, len(ch), ch, 0, nframes, len(o),
                                                o.ctypes.data_as(f32p), a.ctypes.data_as(f32p), float(max_seconds), secs)
        assert done >= 0
        return buf, np.array(secs[:done])


def channels_from_table(tab):
    """fixture table rows (f_r, f_g, f_b, f_c, bit_size, tfloat) -> constructor-order tuples
    (bit_size, f_r, f_g, f_b, f_c, tfloat) as in render.Channel (render.cpp:127)."""
    return [(int(r[4]), float(r[0]), float(r[1]), float(r[2]), float(r[3]), bool(r[5])) for r in tab]


def make_channels(channels):
    """channels: iterable of (bit_size, f_r, f_g, f_b[, f_c[, tfloat]])"""
    arr = (Channel * len(channels))()
    bits = 0
    for i, c in enumerate(channels):
        c = list(c)
        bs, f_r, f_g, f_b = c[:4]
        f_c = c[4] if len(c) > 4 else 0.0
        tf = c[5] if len(c) > 5 else 0
        arr[i].f_r, arr[i].f_g, arr[i].f_b, arr[i].f_c = float(f_r), float(f_g), float(f_b), float(f_c)
        arr[i].bit_size = int(bs)
        arr[i].tfloat = int(tf)
        bits += int(bs)
    return arr, (bits + 7) // 8


def pack_pixel(rgb, channels, reversed_=False):
    ch, bpp = make_channels(channels)
    rgb = _f(rgb)
    out = (C.c_uint8 * bpp)()
    lib().nto_pack_pixel(rgb.ctypes.data_as(f32p), len(ch), ch, int(bool(reversed_)), bpp, out)
    return bytes(out)

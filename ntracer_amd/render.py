"""Mirror of the reference's ``ntracer.render`` module for the ray-cast path.

Same names, argument meaning and error behaviour as the reference's C++ module
(src/render.cpp), but every render goes through libntracer_hip.so to hand-written
HIP kernels -- see include/ntracer_hip.h.  The inter-module capsules of the reference are
intentionally absent (there is one module per type here, not one per dimension).

Pickling follows the reference's wire format (render.cpp:1094-1099, 1197-1208, 1482-1660, 1696-1751):
``__reduce__`` returns ``(render._X_unpickle, (dimension, big-endian IEEE-754 floats, ...))``; with
``ntracer_amd.compat.alias_reference_modules()`` pickles written by the reference load here and vice versa.
"""
import ctypes as C
import os
import threading

from . import _lib
from ._lib import LockedError  # noqa: F401  (re-export: render.LockedError)

MAX_BITSIZE = 31          # render.cpp:48
MAX_PIXELSIZE = 16        # render.cpp:50
DEFAULT_SPECULAR_EXP = 8  # render.cpp:44


def _float_buffer_base():
    """`FloatBuffer` of the one-type CPython module built beside libntracer_hip.so (csrc/nt_pybuffer.c): the buffer protocol
    for Color and Vector, as the reference's types have it.  Without the module they are plain objects."""
    try:
        from . import _pybuffer
        return _pybuffer.FloatBuffer
    except ImportError:
        return object


FloatBuffer = _float_buffer_base()


class Channel(object):
    """render.Channel(bit_size,f_r,f_g,f_b[,f_c=0,tfloat=False]) -- render.cpp:95-164."""
    __slots__ = ("_v",)

    def __init__(self, bit_size, f_r, f_g, f_b, f_c=0, tfloat=False):
        bit_size = int(bit_size)
        tfloat = bool(tfloat)
        if tfloat:
            if bit_size != 32:
                raise ValueError('if "tfloat" is true, "bit_size" can only be 32')
        else:
            if bit_size > MAX_BITSIZE:
                raise ValueError('"bit_size" cannot be greater than %d (unless "tfloat" is true)' % MAX_BITSIZE)
            if bit_size < 1:
                raise ValueError('"bit_size" cannot be less than 1')
        object.__setattr__(self, "_v", (bit_size, float(f_r), float(f_g), float(f_b), float(f_c), tfloat))

    bit_size = property(lambda s: s._v[0])
    f_r = property(lambda s: C.c_float(s._v[1]).value)
    f_g = property(lambda s: C.c_float(s._v[2]).value)
    f_b = property(lambda s: C.c_float(s._v[3]).value)
    f_c = property(lambda s: C.c_float(s._v[4]).value)
    tfloat = property(lambda s: s._v[5])

    def __setattr__(self, k, v):
        raise AttributeError("readonly attribute")

    def __repr__(self):
        return "Channel(%d,%r,%r,%r,%r,%r)" % self._v


class ImageFormat(object):
    """render.ImageFormat(width,height,channels[,pitch=0,reversed=False]) -- render.cpp:167-288."""

    def __init__(self, width, height, channels, pitch=0, reversed=False):
        self.width = int(width)
        self.height = int(height)
        self.pitch = int(pitch)
        self.reversed = bool(reversed)
        self.set_channels(channels)
        if self.width < 1 or self.height < 1:
            raise ValueError("width and height must be at least 1")
        if self.pitch < 0:
            raise ValueError("pitch cannot be negative")
        if self.pitch:
            if self.pitch < self.width * self._bpp:
                raise ValueError('"pitch" must be at least "width" times the size of one pixel in bytes')
        else:
            self.pitch = self.width * self._bpp

    def set_channels(self, channels):
        chans = []
        bits = 0
        for c in channels:
            if not isinstance(c, Channel):
                raise TypeError("object is not an instance of Channel")
            bits += c.bit_size
            chans.append(c)
        if bits > MAX_PIXELSIZE * 8:
            raise ValueError("Too many bytes per pixel. The maximum is %d." % MAX_PIXELSIZE)
        self._channels = tuple(chans)
        self._bpp = (bits + 7) // 8

    @property
    def channels(self):
        return self._channels

    @property
    def bytes_per_pixel(self):
        return self._bpp

    def _as_struct(self):
        arr = (_lib.NtChannel * max(len(self._channels), 1))()
        for i, c in enumerate(self._channels):
            arr[i].f_r, arr[i].f_g, arr[i].f_b, arr[i].f_c = c._v[1:5]
            arr[i].bit_size = c.bit_size
            arr[i].tfloat = 1 if c.tfloat else 0
        f = _lib.NtImageFormat(self.width, self.height, self.pitch, len(self._channels), arr, 1 if self.reversed else 0)
        f._keep = arr
        return f


class Color(FloatBuffer):
    """render.Color(r,g,b) -- light.hpp:4-110, render.cpp:969-1152.  memoryview(c) gives its three floats."""
    __slots__ = ("r", "g", "b")

    def _float_buffer(self):
        import numpy as np
        a = np.array((self.r, self.g, self.b), np.float32)
        a.flags.writeable = False
        return a

    def __init__(self, r, g, b):
        object.__setattr__(self, "r", C.c_float(r).value)
        object.__setattr__(self, "g", C.c_float(g).value)
        object.__setattr__(self, "b", C.c_float(b).value)

    def __setattr__(self, k, v):
        raise AttributeError("readonly attribute")

    @staticmethod
    def _coerce(v):
        if isinstance(v, Color):
            return v
        r, g, b = v
        return Color(r, g, b)

    def __iter__(self):
        return iter((self.r, self.g, self.b))

    def __reduce__(self):
        return _color_unpickle, (_encode_floats((self.r, self.g, self.b)),)

    def __len__(self):
        return 3

    def __getitem__(self, i):
        return (self.r, self.g, self.b)[i]

    def __eq__(self, o):
        try:
            o = Color._coerce(o)
        except (TypeError, ValueError):
            return NotImplemented
        return (self.r, self.g, self.b) == (o.r, o.g, o.b)

    def __ne__(self, o):
        r = self.__eq__(o)
        return r if r is NotImplemented else not r

    def __hash__(self):
        return hash((self.r, self.g, self.b))

    def __add__(self, o):
        o = Color._coerce(o)
        return Color(self.r + o.r, self.g + o.g, self.b + o.b)

    def __sub__(self, o):
        o = Color._coerce(o)
        return Color(self.r - o.r, self.g - o.g, self.b - o.b)

    def __neg__(self):
        return Color(-self.r, -self.g, -self.b)

    def __mul__(self, o):
        if isinstance(o, (int, float)):
            return Color(self.r * o, self.g * o, self.b * o)
        o = Color._coerce(o)
        return Color(self.r * o.r, self.g * o.g, self.b * o.b)

    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, (int, float)):
            return Color(self.r / o, self.g / o, self.b / o)
        o = Color._coerce(o)
        return Color(self.r / o.r, self.g / o.g, self.b / o.b)

    def apply(self, f):
        return Color(f(self.r), f(self.g), f(self.b))

    def __repr__(self):
        return "Color(%r,%r,%r)" % (self.r, self.g, self.b)


class Material(object):
    """render.Material(color[,opacity=1,reflectivity=0,specular_intensity=1,specular_exp=8,
    specular_color=(1,1,1)]) -- render.hpp:56-73, render.cpp:1166-1323."""

    def __init__(self, color, opacity=1, reflectivity=0, specular_intensity=1, specular_exp=DEFAULT_SPECULAR_EXP,
                 specular_color=(1, 1, 1)):
        self.color = Color._coerce(color)
        self.opacity = self._unit(opacity, "opacity")
        self.reflectivity = self._unit(reflectivity, "reflectivity")
        self.specular_intensity = self._unit(specular_intensity, "specular_intensity")
        self.specular_exp = C.c_float(specular_exp).value
        self.specular = Color._coerce(specular_color)

    @staticmethod
    def _unit(v, name):
        v = C.c_float(v).value
        if v < 0 or v > 1:
            raise ValueError("%s must be between 0 and 1" % name)
        return v

    def _key(self):
        return (tuple(self.color), tuple(self.specular), self.opacity, self.reflectivity, self.specular_intensity,
                self.specular_exp)

    def __eq__(self, o):
        return self._key() == o._key() if isinstance(o, Material) else NotImplemented

    def __ne__(self, o):
        r = self.__eq__(o)
        return r if r is NotImplemented else not r

    def __hash__(self):
        return hash(self._key())

    def __reduce__(self):
        return _material_unpickle, (_encode_floats(tuple(self.color) + tuple(self.specular) + (
            self.opacity, self.reflectivity, self.specular_intensity, self.specular_exp)),)

    def _as_struct(self):
        m = _lib.NtMaterial()
        m.color[:] = tuple(self.color)
        m.specular[:] = tuple(self.specular)
        m.opacity = self.opacity
        m.reflectivity = self.reflectivity
        m.specular_intensity = self.specular_intensity
        m.specular_exp = self.specular_exp
        return m

    def __repr__(self):
        return "Material(%r,%r,%r,%r,%r,%r)" % (tuple(self.color), self.opacity, self.reflectivity,
                                                self.specular_intensity, self.specular_exp, tuple(self.specular))


# ---- pickling: render.cpp:1400-1660.  Floats travel as big-endian IEEE-754 (encode_float_ieee754, :1400-1437).

def _encode_floats(values):
    import numpy as np
    return np.asarray(values, ">f4").tobytes()


def _decode_floats(data, count, what):
    import numpy as np
    if not isinstance(data, (bytes, bytearray)):
        raise TypeError("object is not an instance of bytes")
    if len(data) != 4 * count:
        raise ValueError("%s data is malformed" % what)
    return np.frombuffer(bytes(data), ">f4").astype(np.float32)


def _dimension(d):
    d = int(d)
    if d < 3:
        raise ValueError("dimension cannot be less than 3")      # get_dimension, render.cpp:1385-1389
    return d


def _color_unpickle(data):
    v = _decode_floats(data, 3, "color")
    return Color(float(v[0]), float(v[1]), float(v[2]))


def _material_unpickle(data):
    v = [float(x) for x in _decode_floats(data, 10, "material")]
    m = Material.__new__(Material)         # the reference assigns the fields without range checks (:1502-1512)
    m.color = Color(*v[0:3])
    m.specular = Color(*v[3:6])
    m.opacity, m.reflectivity, m.specular_intensity, m.specular_exp = v[6:10]
    return m


def _vector_unpickle(dim, data):
    from . import tracern
    n = _dimension(dim)
    return tracern.Vector._wrap(_decode_floats(data, n, "vector"))


def _matrix_unpickle(dim, data):
    from . import tracern
    n = _dimension(dim)
    return tracern.Matrix._wrap(_decode_floats(data, n * n, "matrix").reshape(n, n))


def _triangle_unpickle(dim, data, material):
    from . import tracern
    n = _dimension(dim)
    v = _decode_floats(data, n * (n + 1), "triangle").reshape(n + 1, n)      # p1, face_normal, edge normals
    if not isinstance(material, Material):
        raise TypeError("object is not an instance of Material")
    return tracern.Triangle(v[0], v[1], v[2:], material)


def _triangle_batch_unpickle(batch_size, dim, data, *materials):
    from . import tracern
    n = _dimension(dim)
    if int(batch_size) != tracern.BATCH_SIZE:
        raise TypeError("The TriangleBatch instance was pickled with a different batch size. It cannot be loaded here.")
    if len(materials) != tracern.BATCH_SIZE:
        raise TypeError("wrong number of arguments")
    # rows p1, face_normal, edge normals; each row [component][lane] (vector<Store,v_real>, tracer.hpp:532-548)
    v = _decode_floats(data, tracern.BATCH_SIZE * n * (n + 1), "triangle batch").reshape(n + 1, n, tracern.BATCH_SIZE)
    for m in materials:
        if not isinstance(m, Material):
            raise TypeError("object is not an instance of Material")
    return tracern.TriangleBatch([tracern.Triangle(v[0, :, l], v[1, :, l], v[2:, :, l], materials[l])
                                  for l in range(tracern.BATCH_SIZE)])


def _solid_unpickle(dim, data, material):
    from . import tracern
    n = _dimension(dim)
    if not isinstance(data, (bytes, bytearray)):
        raise TypeError("object is not an instance of bytes")
    if len(data) != 4 * n * (n + 1) + 1:
        raise ValueError("solid data is malformed")
    if data[0] not in (1, 2):
        raise ValueError("solid data is corrupt")
    if not isinstance(material, Material):
        raise TypeError("object is not an instance of Material")
    v = _decode_floats(bytes(data[1:]), n * (n + 1), "solid")
    return tracern.Solid(int(data[0]), v[n * n:], tracern.Matrix._wrap(v[:n * n].reshape(n, n)), material)


def _aabb_unpickle(dim, data):
    from . import tracern
    n = _dimension(dim)
    v = _decode_floats(data, 2 * n, "AABB")
    return tracern.AABB(n, v[:n], v[n:])


class Scene(object):
    """render.Scene: the plugin interface `class scene` (render.hpp:8-26) as seen from Python.
    Concrete scenes (tracern.BoxScene / CompositeScene) own an nt_scene_t handle."""
    _handle = None

    def __new__(cls, *a, **k):
        if cls is Scene:
            raise TypeError("the Scene type cannot be instantiated directly")
        return object.__new__(cls)

    def __del__(self):
        h, self._handle = self._handle, None
        if h:
            try:
                _lib.lib().nt_scene_destroy(h)
            except Exception:
                pass

    def calculate_color(self, x, y, width, height):
        """Scene.calculate_color(x,y,width,height) -- render.cpp:586-614."""
        rgb = (C.c_float * 3)()
        _lib.check(_lib.lib().nt_calculate_color(self._handle, int(x), int(y), int(width), int(height), rgb))
        return Color(rgb[0], rgb[1], rgb[2])

    @property
    def locked(self):
        return bool(_lib.lib().nt_scene_locked(self._handle))


def _device_pointer(dest):
    """(ptr, nbytes, device_index, stream) for a torch CUDA/HIP tensor, else None."""
    if type(dest).__module__.split(".")[0] != "torch":
        return None
    if not getattr(dest, "is_cuda", False):
        return None
    import torch
    if not dest.is_contiguous():
        raise ValueError("device destination must be contiguous")
    stream = torch.cuda.current_stream(dest.device).cuda_stream
    return dest.data_ptr(), dest.numel() * dest.element_size(), dest.device.index, stream


def _host_buffer(dest):
    try:
        mv = memoryview(dest)
    except TypeError:
        raise TypeError("dest must support the buffer protocol")
    if mv.readonly:
        raise BufferError("Object is not writable.")
    if not mv.c_contiguous:
        raise BufferError("dest must be C-contiguous")
    n = mv.nbytes
    arr = (C.c_char * n).from_buffer(dest) if n else None
    return arr, n


def _opts(device=-1, band_rank=0, band_world=1, compact=False, collect_stats=False, band_rows=0, strict_reference=None, overlapped=False):
    o = _lib.NtRenderOpts()
    o.overlapped = 1 if overlapped else 0
    o.device = device
    o.band_rank = band_rank
    o.band_world = band_world
    o.band_rows = band_rows
    o.compact = 1 if compact else 0
    # None: follow NTRACER_STRICT_REFERENCE (default 0 = skip k-d cells beyond the current hit; same pixels)
    if strict_reference is None:
        strict_reference = os.environ.get("NTRACER_STRICT_REFERENCE", "0") not in ("", "0")
    o.strict_reference = 1 if strict_reference else 0
    o.collect_stats = 1 if collect_stats else 0
    return o


class CameraTable(object):
    """A camera path resident in device memory (``nt_camera_table_create``): the cameras of a sequence -- e.g. the 160 of
    the reference's RotatingCamera loop, scripts/polytope.py:522-556 -- packed and uploaded once.  ``render(scene, dest,
    format)`` then renders one frame per camera into ``dest`` (a torch HIP tensor of at least ``frames * frame_bytes``
    bytes; the launch is enqueued on torch's current stream) with nothing to pack or upload per call."""

    def __init__(self, dimension, origins, axes, device=-1):
        import numpy as np
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, dimension)
        a = np.ascontiguousarray(axes, np.float32).reshape(len(o), dimension, dimension)
        self.dimension = int(dimension)
        self.frames = len(o)
        self._h = _lib.lib().nt_camera_table_create(self.dimension, self.frames, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), int(device))
        if not self._h:
            raise RuntimeError(_lib.last_error())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().nt_camera_table_destroy(h)
            except Exception:
                pass

    def render(self, scene, dest, format, frame_bytes=None, band_rank=0, band_world=1, compact=False, band_rows=0, strict_reference=None,
               overlapped=False, first=0, count=None):
        """``overlapped``: the caller keeps two or more torch streams busy with calls like this one (nt_render_opts.overlapped:
        the launches are shaped for throughput rather than for the time of a call that runs alone; same pixels)."""
        dev = _device_pointer(dest)
        if dev is None:
            raise TypeError("dest must be a torch HIP tensor")
        ptr, nbytes, index, stream = dev
        fmt = format._as_struct()
        opts = _opts(index, band_rank, band_world, compact, False, band_rows=band_rows, strict_reference=strict_reference, overlapped=overlapped)
        if count is None:
            count = self.frames - first                 # (frames [first, first + count) of the table go to dest's frames 0 .. count-1)
        if frame_bytes is None:
            frame_bytes = nbytes // max(count, 1)
        _lib.check(_lib.lib().nt_render_table_device(scene._handle, C.c_void_p(ptr), frame_bytes, self._h, int(first), int(count), C.byref(fmt),
                                                     C.byref(opts), C.c_void_p(stream)))
        return True


class BlockingRenderer(object):
    """render.BlockingRenderer([threads=-1]) -- render.cpp:769-923.

    ``threads`` is accepted for compatibility; the frame is rendered by one HIP launch on
    ``device`` (default: current device).  ``render`` returns True, or False if
    ``signal_abort`` was called from another thread while it ran."""

    def __init__(self, threads=-1, device=-1):
        self.threads = int(threads)
        self.device = int(device)
        self._abort = C.c_int(0)
        self._mut = threading.Lock()
        self._busy = False

    def render(self, dest, format, scene, band_rank=0, band_world=1, compact=False, collect_stats=False, strict_reference=None,
               band_rows=0, overlapped=False):
        if not isinstance(format, ImageFormat):
            raise TypeError("format must be an ImageFormat")
        if not isinstance(scene, Scene):
            raise TypeError("scene must be a Scene")
        with self._mut:
            if self._busy:
                raise RuntimeError("the renderer is already running")
            self._busy = True
            self._abort.value = 0
        try:
            fmt = format._as_struct()
            dev = _device_pointer(dest)
            L = _lib.lib()
            if dev is not None:
                ptr, nbytes, index, stream = dev
                # (overlapped: device destinations only -- the caller keeps several torch streams busy, nt_render_opts.overlapped)
                opts = _opts(index, band_rank, band_world, compact, collect_stats, band_rows=band_rows, strict_reference=strict_reference,
                             overlapped=overlapped)
                _lib.check(L.nt_render_device(scene._handle, C.c_void_p(ptr), nbytes, C.byref(fmt), C.byref(opts),
                                              C.c_void_p(stream)))
                return True
            arr, n = _host_buffer(dest)
            opts = _opts(self.device, band_rank, band_world, compact, collect_stats, band_rows=band_rows, strict_reference=strict_reference)
            r = _lib.check(L.nt_render(scene._handle, arr, n, C.byref(fmt), C.byref(opts), C.byref(self._abort)))
            return r != _lib.NT_ABORTED
        finally:
            with self._mut:
                self._busy = False

    def signal_abort(self):
        self._abort.value = 1


class CallbackRenderer(object):
    """render.CallbackRenderer([threads=0]) -- render.cpp:495-766: asynchronous render, ``callback(renderer)``
    is invoked from a worker thread on completion (not after ``abort_render``)."""

    def __init__(self, threads=0, device=-1):
        self.threads = int(threads)
        self.device = int(device)
        self._abort = C.c_int(0)
        self._mut = threading.Lock()
        self._worker = None

    def begin_render(self, dest, format, scene, callback):
        if not isinstance(format, ImageFormat):
            raise TypeError("format must be an ImageFormat")
        if not isinstance(scene, Scene):
            raise TypeError("scene must be a Scene")
        arr, n = _host_buffer(dest)
        fmt = format._as_struct()
        if fmt.pitch * fmt.height > n:
            raise ValueError("the buffer is too small for an image with the given dimensions")
        with self._mut:
            if self._worker is not None and self._worker.is_alive():
                raise RuntimeError("the renderer is already running")
            self._abort.value = 0
            L = _lib.lib()
            _lib.check(L.nt_scene_lock(scene._handle))     # sc.lock() before returning (render.cpp:688)
            opts = _opts(self.device)

            def work():
                try:
                    r = L.nt_render(scene._handle, arr, n, C.byref(fmt), C.byref(opts), C.byref(self._abort))
                finally:
                    L.nt_scene_unlock(scene._handle)
                if r == _lib.NT_OK:
                    try:
                        callback(self)
                    except Exception:                      # the reference prints and carries on (render.cpp:536-542)
                        import traceback
                        traceback.print_exc()
                elif r < 0:
                    import sys
                    sys.stderr.write("error: %s\n" % _lib.last_error())

            self._worker = threading.Thread(target=work, daemon=True)
            self._worker.start()

    def abort_render(self):
        w = self._worker
        if w is not None and w.is_alive():
            self._abort.value = 1
            w.join()
        self._abort.value = 0


def get_optimized_tracern(dimension):
    """render.get_optimized_tracern(dimension) -- render.cpp:1659-1674.  The specialisation for
    3..10 dimensions happens inside the HIP library (template<int N> kernels); the Python surface is
    one module."""
    from . import tracern
    return tracern

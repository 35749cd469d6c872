"""Mirror of the reference's ``ntracer.tracern`` module for the ray-cast path
(src/ntracer_body.hpp): BoxScene, CompositeScene, Camera, Vector, Matrix, AABB, the k-d node
and primitive value types and the lights.  Scenes own a handle of the HIP library; everything
else is a light fp32 value type used to describe a scene.

Scene construction (Triangle.from_points/to_points, prototypes, build_kdtree, build_composite_scene) lives in
``builder.py``: SURVEY section 8f item 1.  Not mirrored: pickling, TriangleBatchPointData-style introspection.
"""
import ctypes as C
import math

import numpy as np

from . import _lib, builder
from . import render as _render
from .render import Color, Material, Scene

BATCH_SIZE = _lib.NT_BATCH_SIZE     # tracer.hpp:34-38 (SSE reference build)
CUBE, SPHERE = 1, 2                  # wrapper.CUBE / wrapper.SPHERE (tracer.hpp:225)

f32 = np.float32


def _vec(n, values):
    if isinstance(values, Vector):
        if values.dimension != n:
            raise TypeError("vector has the wrong dimension")
        return values._v
    a = np.asarray(list(values), dtype=f32)
    if a.shape != (n,):
        raise TypeError("expected %d values" % n)
    return a


class Vector(_render.FloatBuffer):
    """tracern.Vector(dimension[,values]) -- fp32 n-vector (geometry.hpp:131-283).  memoryview(v) gives its floats."""
    __slots__ = ("_v",)

    def _float_buffer(self):
        a = self._v.view()
        a.flags.writeable = False
        return a

    def __init__(self, dimension, values=None):
        dimension = int(dimension)
        if dimension < 1:
            raise ValueError("dimension must be positive")
        self._v = np.zeros(dimension, f32) if values is None else _vec(dimension, values).copy()

    @classmethod
    def _wrap(cls, a):
        v = cls.__new__(cls)
        v._v = np.ascontiguousarray(a, dtype=f32)
        return v

    @staticmethod
    def axis(dimension, axis, length=1):
        v = np.zeros(int(dimension), f32)
        v[axis] = length
        return Vector._wrap(v)

    dimension = property(lambda s: len(s._v))

    def __reduce__(self):                               # obj_Vector_reduce, render.cpp:1705-1710
        return _render._vector_unpickle, (len(self._v), _render._encode_floats(self._v))

    def __len__(self):
        return len(self._v)

    def __getitem__(self, i):
        return float(self._v[i])

    def __iter__(self):
        return (float(x) for x in self._v)

    def __add__(self, o):
        return Vector._wrap(self._v + _vec(len(self._v), o))

    def __sub__(self, o):
        return Vector._wrap(self._v - _vec(len(self._v), o))

    def __neg__(self):
        return Vector._wrap(-self._v)

    def __mul__(self, s):
        return Vector._wrap(self._v * f32(s))

    __rmul__ = __mul__

    def __truediv__(self, s):
        return Vector._wrap(self._v / f32(s))

    def __eq__(self, o):
        try:
            return bool(np.array_equal(self._v, _vec(len(self._v), o)))
        except TypeError:
            return NotImplemented

    def __ne__(self, o):
        r = self.__eq__(o)
        return r if r is NotImplemented else not r

    def set_c(self, index, value):
        """Vector.set_c(index,value) -> a copy with one component replaced (ntracer_body.hpp:1998-2014)"""
        index = int(index)
        if index < 0 or index >= len(self._v):
            raise IndexError("vector index out of range")
        r = self._v.copy()
        r[index] = f32(value)
        return Vector._wrap(r)

    def square(self):
        return float(dot(self, self))

    def absolute(self):
        return float(np.sqrt(f32(dot(self, self))))

    def unit(self):
        return Vector._wrap(self._v / np.sqrt(f32(dot(self, self))))

    def apply(self, f):
        return Vector._wrap([f(float(x)) for x in self._v])

    def __repr__(self):
        return "Vector(%d,%r)" % (len(self._v), [float(x) for x in self._v])


def dot(a, b):
    """tracern.dot(a,b) -- fp32, summed left to right."""
    av = a._v if isinstance(a, Vector) else np.asarray(list(a), f32)
    bv = b._v if isinstance(b, Vector) else np.asarray(list(b), f32)
    if av.shape != bv.shape:
        raise TypeError("cannot perform dot product on vectors of different dimension")
    s = f32(av[0] * bv[0])
    for k in range(1, len(av)):
        s = f32(s + f32(av[k] * bv[k]))
    return float(s)


class Matrix(object):
    """tracern.Matrix(dimension,values) -- fp32 n x n (geometry.hpp:527-844); only what cameras and
    Solid orientations need."""
    __slots__ = ("_m",)

    def __init__(self, dimension, values):
        n = int(dimension)
        a = np.asarray([list(r) if not isinstance(r, (int, float)) else r for r in values], dtype=f32)
        if a.size != n * n:
            raise TypeError("expected %d values" % (n * n))
        self._m = a.reshape(n, n).copy()

    @classmethod
    def _wrap(cls, a):
        m = object.__new__(cls)
        m._m = np.ascontiguousarray(a, dtype=f32)
        return m

    dimension = property(lambda s: s._m.shape[0])

    def __reduce__(self):                               # obj_Matrix_reduce, render.cpp:1711-1715
        return _render._matrix_unpickle, (self._m.shape[0], _render._encode_floats(self._m.ravel()))

    @staticmethod
    def identity(dimension):
        return Matrix._wrap(np.eye(int(dimension), dtype=f32))

    @staticmethod
    def scale(dimension, s=None):
        n = int(dimension)
        if isinstance(s, (int, float)):
            return Matrix._wrap(np.eye(n, dtype=f32) * f32(s))
        return Matrix._wrap(np.diag(_vec(n, s)))

    @staticmethod
    def rotation(a, b, theta):
        """Matrix.rotation(a,b,theta): rotation_ (geometry.hpp:579-591), fp32."""
        av, bv = a._v, b._v
        n = len(av)
        c = f32(f32(math.cos(f32(theta))) - f32(1))
        s = f32(math.sin(f32(theta)))
        m = np.zeros((n, n), f32)
        for row in range(n):
            for col in range(n):
                x = f32(f32(av[row] * f32(f32(av[col] * c) - f32(bv[col] * s))) + f32(bv[row] * f32(f32(bv[col] * c) + f32(av[col] * s))))
                if col == row:
                    x = f32(x + f32(1))
                m[row, col] = x
        return Matrix._wrap(m)

    @staticmethod
    def reflection(a):
        """Matrix.reflection(a): I - 2 a a^T / |a|^2 (geometry.hpp:602-608), fp32."""
        av = a._v if isinstance(a, Vector) else np.asarray(list(a), f32)
        n = len(av)
        square = f32(dot(Vector._wrap(av), Vector._wrap(av)))
        m = np.zeros((n, n), f32)
        for row in range(n):
            for col in range(n):
                m[row, col] = f32(f32(1 if row == col else 0) - f32(f32(f32(f32(2) * av[row]) * av[col]) / square))
        return Matrix._wrap(m)

    def __getitem__(self, i):
        return Vector._wrap(self._m[i].copy())

    def __len__(self):
        return self._m.shape[0]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def determinant(self):
        """Matrix.determinant() (ntracer_body.hpp:2326-2330; the reference runs an fp32 LU, geometry.hpp:790-823)"""
        return float(f32(np.linalg.det(self._m.astype(np.float64))))

    @property
    def values(self):
        """all elements, row-major (an attribute in the reference: ntracer_body.hpp:2330-2345)"""
        return tuple(float(x) for x in self._m.ravel())

    def __mul__(self, o):
        if isinstance(o, Matrix):
            return Matrix._wrap(_matmul(self._m, o._m.T))
        if isinstance(o, Vector):
            return Vector._wrap([dot(Vector._wrap(r), o) for r in self._m])
        return NotImplemented

    def transpose(self):
        return Matrix._wrap(self._m.T.copy())

    def inverse(self):
        inv = np.linalg.inv(self._m.astype(np.float64))
        return Matrix._wrap(inv.astype(f32))

    def __eq__(self, o):
        return isinstance(o, Matrix) and bool(np.array_equal(self._m, o._m))

    def __repr__(self):
        return "Matrix(%d,%r)" % (len(self), list(self.values))


def _matmul(a, bt):
    """r[row][col] = dot(a[row], bt[col]) in fp32, left to right (geometry.hpp:556-575)."""
    n = a.shape[0]
    r = np.zeros((n, n), f32)
    for row in range(n):
        for col in range(n):
            r[row, col] = dot(Vector._wrap(a[row]), Vector._wrap(bt[col]))
    return r


class CameraAxes(object):
    def __init__(self, cam):
        self._cam = cam

    def __len__(self):
        return self._cam.dimension

    def __getitem__(self, i):
        return Vector._wrap(self._cam._axes[i].copy())

    def __setitem__(self, i, v):
        self._cam._axes[i] = _vec(self._cam.dimension, v)


class Camera(object):
    """tracern.Camera(dimension) -- camera.hpp:7-45: origin + orientation rows (right, up, forward, ...)."""

    def __init__(self, dimension):
        n = int(dimension)
        if n < 3:
            raise ValueError("dimension cannot be less than 3")
        self._origin = np.zeros(n, f32)
        self._axes = np.eye(n, dtype=f32)

    dimension = property(lambda s: len(s._origin))

    @property
    def origin(self):
        return Vector._wrap(self._origin.copy())

    @origin.setter
    def origin(self, v):
        self._origin = _vec(self.dimension, v).copy()

    @property
    def axes(self):
        return CameraAxes(self)

    def translate(self, v):
        """origin += v[i]*axes[i] (camera.hpp:17-19)."""
        v = _vec(self.dimension, v)
        for i in range(self.dimension):
            self._origin = self._origin + v[i] * self._axes[i]

    def transform(self, m):
        """t_orientation = t_orientation.mult_transpose(m) (camera.hpp:21-23)."""
        self._axes = _matmul(self._axes, m._m)

    def normalize(self):
        """Gram-Schmidt as camera.hpp:25-36 (fp32, same association order)."""
        n = self.dimension
        ax = self._axes
        new_axes = []
        for i in range(n - 1):
            x = np.zeros(n, f32)
            for j in range(i):
                x = x + f32(dot(Vector._wrap(ax[i + 1]), Vector._wrap(ax[j]))) * ax[j]
            new_axes.append(ax[i + 1] - x)
        out = np.zeros((n, n), f32)
        out[0] = ax[0] / np.sqrt(f32(dot(Vector._wrap(ax[0]), Vector._wrap(ax[0]))))
        for i in range(1, n):
            v = new_axes[i - 1]
            out[i] = v / np.sqrt(f32(dot(Vector._wrap(v), Vector._wrap(v))))
        self._axes = out

    def _copy(self):
        c = Camera(self.dimension)
        c._origin = self._origin.copy()
        c._axes = self._axes.copy()
        return c


class AABB(object):
    """tracern.AABB(dimension[,start,end]) -- tracer.hpp:1327-1356."""

    def __init__(self, dimension, start=None, end=None):
        n = int(dimension)
        self.dimension = n
        self.start = Vector(n, start) if start is not None else Vector._wrap(np.full(n, np.finfo(f32).min, f32))
        self.end = Vector(n, end) if end is not None else Vector._wrap(np.full(n, np.finfo(f32).max, f32))

    def __reduce__(self):                               # obj_AABB_reduce, render.cpp:1745-1752
        return _render._aabb_unpickle, (self.dimension, _render._encode_floats(np.concatenate([self.start._v, self.end._v])))

    def __eq__(self, o):
        if not isinstance(o, AABB):
            return NotImplemented
        return self.start == o.start and self.end == o.end

    def __ne__(self, o):
        r = self.__eq__(o)
        return r if r is NotImplemented else not r

    __hash__ = None

    def _check_axis(self, axis):
        axis = int(axis)
        if axis < 0 or axis >= self.dimension:
            raise IndexError("index out of range")
        return axis

    def left(self, axis, split):
        """AABB.left(axis,split): the part below the split (ntracer_body.hpp:2498-2513, tracer.hpp:1337-1343)."""
        axis = self._check_axis(axis)
        split = float(split)
        if not (self.start[axis] < split < self.end[axis]):
            raise ValueError("\"split\" must be inside the box within the given axis")
        e = self.end._v.copy()
        e[axis] = split
        return AABB(self.dimension, self.start._v, e)

    def right(self, axis, split):
        """AABB.right(axis,split): the part above the split."""
        axis = self._check_axis(axis)
        split = float(split)
        if not (self.start[axis] < split < self.end[axis]):
            raise ValueError("\"split\" must be inside the box within the given axis")
        b = self.start._v.copy()
        b[axis] = split
        return AABB(self.dimension, b, self.end._v)

    def intersects(self, primitive):
        """AABB.intersects(prototype) -> bool: whether the box and the primitive share a region of non-zero extent
        (two things that merely touch do not intersect; tracer.hpp:1459-1463).  The reference answers with
        projection tests (:1465-1700); here the primitive is clipped to the box exactly (nt_polytope_clip_box):
        a simplex as a polytope in its own plane, a cube solid as a parallelotope; a sphere solid by the distance
        from its centre to the box in the solid's own coordinates."""
        if not isinstance(primitive, PrimitivePrototype):
            raise TypeError("object is not an instance of PrimitivePrototype")
        if primitive.dimension != self.dimension:
            raise TypeError("cannot perform intersection test on object with different dimension")
        return primitive._overlaps(self.start._v.astype(np.float64), self.end._v.astype(np.float64))


class PointLight(object):
    """tracern.PointLight(position,color) -- tracer.hpp:1678-1689."""

    def __init__(self, position, color):
        self.position = position if isinstance(position, Vector) else Vector(len(list(position)), position)
        self.color = Color._coerce(color)

    dimension = property(lambda s: s.position.dimension)


class GlobalLight(object):
    """tracern.GlobalLight(direction,color) -- tracer.hpp:1691-1698."""

    def __init__(self, direction, color):
        self.direction = direction if isinstance(direction, Vector) else Vector(len(list(direction)), direction)
        self.color = Color._coerce(color)

    dimension = property(lambda s: s.direction.dimension)


# ---------------------------------------------------------------------------------------------
# primitives and k-d nodes: plain descriptions, flattened into nt_scene_desc by CompositeScene
# ---------------------------------------------------------------------------------------------
class Primitive(object):
    pass


class PrimitiveBatch(object):
    pass


class Triangle(Primitive):
    """tracern.Triangle(p1,face_normal,edge_normals,material) -- an (n-1)-simplex in plane /
    edge-normal form (tracer.hpp:392-488)."""

    def __init__(self, p1, face_normal, edge_normals, material):
        p1 = list(p1)
        n = len(p1)
        self.p1 = Vector(n, p1)
        self.face_normal = Vector(n, face_normal)
        self.edge_normals = tuple(Vector(n, e) for e in edge_normals)
        if len(self.edge_normals) != n - 1:
            raise ValueError("a simplex of dimension %d needs %d edge normals" % (n, n - 1))
        if not isinstance(material, Material):
            raise TypeError("material must be a Material")
        self.material = material
        self.d = -dot(self.face_normal, self.p1)       # recalculate_d (tracer.hpp:472-474)

    dimension = property(lambda s: s.p1.dimension)

    @staticmethod
    def from_points(points, material):
        """Triangle.from_points(points,material) -- tracer.hpp:442-462."""
        p1, fn, edges = builder.from_points_record([list(p) for p in points])
        return Triangle(p1, fn, edges, material)

    def to_points(self):
        """Triangle.to_points() -- tracer.hpp:490-506."""
        pts = builder.to_points_array(self.p1._v, self.face_normal._v, [e._v for e in self.edge_normals])
        return tuple(Vector._wrap(p) for p in pts)

    def _rows(self):
        return np.vstack([self.p1._v, self.face_normal._v] + [e._v for e in self.edge_normals])      # [n+1][n]

    def __eq__(self, o):
        if not isinstance(o, Triangle):
            return NotImplemented
        return self.dimension == o.dimension and np.array_equal(self._rows(), o._rows()) and self.material == o.material

    def __ne__(self, o):
        r = self.__eq__(o)
        return r if r is NotImplemented else not r

    __hash__ = object.__hash__

    def __reduce__(self):                               # obj_Triangle_reduce, ntracer_body.hpp:1217-1233
        return _render._triangle_unpickle, (self.dimension, _render._encode_floats(self._rows().ravel()), self.material)

    def _record(self):
        rec = [f32(self.d)] + list(self.face_normal._v) + list(self.p1._v)
        for e in self.edge_normals:
            rec += list(e._v)
        return np.asarray(rec, f32)


class TriangleBatch(PrimitiveBatch):
    """tracern.TriangleBatch(triangles): exactly BATCH_SIZE simplices (tracer.hpp:532-641)."""

    def __init__(self, triangles):
        tris = tuple(triangles)
        if len(tris) != BATCH_SIZE or not all(isinstance(t, Triangle) for t in tris):
            raise ValueError("exactly %d Triangle instances are required" % BATCH_SIZE)
        if len(set(t.dimension for t in tris)) != 1:
            raise TypeError("the triangles must have the same dimension")
        self._tris = tris

    dimension = property(lambda s: s._tris[0].dimension)

    def __len__(self):
        return BATCH_SIZE

    def __getitem__(self, i):
        return self._tris[i]

    def __reduce__(self):                               # rows [component][lane], render.cpp:1725-1735
        vals = np.stack([t._rows() for t in self._tris], axis=2)          # [n+1][n][BATCH_SIZE]
        return _render._triangle_batch_unpickle, (BATCH_SIZE, self.dimension, _render._encode_floats(vals.ravel())) + tuple(
            t.material for t in self._tris)


class Solid(Primitive):
    """tracern.Solid(type,position,orientation,material) -- tracer.hpp:231-289."""

    def __init__(self, type, position, orientation, material):
        if type not in (CUBE, SPHERE):
            raise ValueError("invalid shape type")
        if not isinstance(orientation, Matrix):
            raise TypeError("orientation must be a Matrix")
        n = orientation.dimension
        self.type = type
        self.orientation = orientation
        self.inv_orientation = orientation.inverse()
        self.position = Vector(n, position)
        if not isinstance(material, Material):
            raise TypeError("material must be a Material")
        self.material = material

    dimension = property(lambda s: s.orientation.dimension)

    def __reduce__(self):                               # obj_Solid_reduce, render.cpp:1736-1744
        data = bytes([self.type]) + _render._encode_floats(self.orientation._m.ravel()) + _render._encode_floats(self.position._v)
        return _render._solid_unpickle, (self.dimension, data, self.material)


class KDNode(object):
    pass


class KDLeaf(KDNode):
    """tracern.KDLeaf(primitives): batches are kept in front (tracer.hpp:1146-1150)."""

    def __init__(self, primitives):
        prims = list(primitives)
        if not prims:
            raise ValueError("KDLeaf requires at least one item")
        for p in prims:
            if not isinstance(p, (Primitive, PrimitiveBatch)):
                raise TypeError("object is not an instance of Primitive or PrimitiveBatch")
        if len(set(p.dimension for p in prims)) != 1:
            raise TypeError("every member of KDLeaf must have the same dimension")
        self._items = tuple([p for p in prims if isinstance(p, PrimitiveBatch)] +
                            [p for p in prims if not isinstance(p, PrimitiveBatch)])

    dimension = property(lambda s: s._items[0].dimension)

    def __len__(self):
        return len(self._items)

    def __getitem__(self, i):
        return self._items[i]


class KDBranch(KDNode):
    """tracern.KDBranch(axis,split[,left=None,right=None]) -- tracer.hpp:813-830."""

    def __init__(self, axis, split, left=None, right=None):
        if left is None and right is None:
            raise ValueError('"left" and "right" can\'t both be None')
        for c in (left, right):
            if c is not None and not isinstance(c, KDNode):
                raise TypeError("child is not a KDNode")
        if left is not None and right is not None and left.dimension != right.dimension:
            raise TypeError("the nodes must have the same dimension")
        self.axis = int(axis)
        self.split = float(f32(split))
        self.left = left
        self.right = right
        if self.axis < 0 or self.axis >= self.dimension:
            raise ValueError("invalid axis")

    dimension = property(lambda s: (s.left if s.left is not None else s.right).dimension)


class _SceneBase(Scene):
    """Camera / fov handling shared by BoxScene and CompositeScene (ntracer_body.hpp:676-715)."""

    def _init_common(self, n):
        self._n = n

    dimension = property(lambda s: s._n)

    @property
    def fov(self):
        return float(_lib.lib().nt_scene_get_fov(self._handle))

    def set_fov(self, fov):
        _lib.check(_lib.lib().nt_scene_set_fov(self._handle, float(fov)))

    def set_camera(self, camera):
        if not isinstance(camera, Camera) or camera.dimension != self._n:
            raise TypeError("the scene and camera must have the same dimension")
        o = np.ascontiguousarray(camera._origin, f32)
        a = np.ascontiguousarray(camera._axes, f32)
        _lib.check(_lib.lib().nt_scene_set_camera(self._handle, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p)))

    def get_camera(self):
        c = Camera(self._n)
        o = np.zeros(self._n, f32)
        a = np.zeros((self._n, self._n), f32)
        _lib.check(_lib.lib().nt_scene_get_camera(self._handle, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p)))
        c._origin, c._axes = o, a
        return c

    def _set_camera_arrays(self, origin, axes):
        o = np.ascontiguousarray(origin, f32)
        a = np.ascontiguousarray(axes, f32).reshape(self._n, self._n)
        _lib.check(_lib.lib().nt_scene_set_camera(self._handle, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p)))

    def colors_at(self, xs, ys, width, height, device=-1):
        """fp32 colours of many pixels in one launch (batched Scene.calculate_color)."""
        xs = np.ascontiguousarray(xs, np.int32)
        ys = np.ascontiguousarray(ys, np.int32)
        out = np.zeros((len(xs), 3), f32)
        _lib.check(_lib.lib().nt_colors_at(self._handle, int(width), int(height), len(xs), xs.ctypes.data_as(_lib.i32p),
                                           ys.ctypes.data_as(_lib.i32p), out.ctypes.data_as(_lib.f32p), int(device)))
        return out

    def last_stats(self):
        st = _lib.NtStats()
        _lib.check(_lib.lib().nt_scene_last_stats(self._handle, C.byref(st)))
        return st.as_dict()


class BoxScene(_SceneBase):
    """tracern.BoxScene(dimension) -- one hypercube [-1,1]^n (tracer.hpp:83-123)."""

    def __init__(self, dimension):
        n = int(dimension)
        h = _lib.lib().nt_box_scene_create(n)
        if not h:
            raise ValueError(_lib.last_error())
        self._handle = h
        self._init_common(n)


_FLAT_KEYS = ("root", "node_axis", "node_split", "node_left", "node_right", "items", "batch_recs", "batch_mats",
              "tri_recs", "tri_mats", "solid_recs", "solid_types", "solid_mats", "materials", "aabb_start", "aabb_end")


class CompositeScene(_SceneBase):
    """tracern.CompositeScene(boundary,data) -- the contents of a k-d tree (tracer.hpp:1710-1927)."""

    def __init__(self, boundary, data):
        if not isinstance(boundary, AABB):
            raise TypeError("boundary must be an AABB")
        if not isinstance(data, KDNode):
            raise TypeError("data must be a KDNode")
        if boundary.dimension != data.dimension:
            raise TypeError("boundary and data must have the same dimension")
        self._create(self._flatten(boundary, data))
        self._root_obj = data

    @classmethod
    def from_flat(cls, dimension, flat):
        """Build directly from flat arrays (the layout of nt_scene_desc / tests/golden/*.npz)."""
        self = object.__new__(cls)
        d = {k: flat[k] for k in _FLAT_KEYS}
        d["dimension"] = int(dimension)
        self._create(d)
        return self

    @staticmethod
    def _flatten(boundary, root):
        n = boundary.dimension
        nodes, items = [], []
        batch_ids, tri_ids, solid_ids, mat_ids = {}, {}, {}, {}
        batch_recs, batch_mats, tri_recs, tri_mats, solid_recs, solid_types, solid_mats, mats = [], [], [], [], [], [], [], []

        def mat(m):
            k = m._key()
            if k not in mat_ids:
                mat_ids[k] = len(mats)
                mats.append(list(m.color) + list(m.specular) + [m.opacity, m.reflectivity, m.specular_intensity, m.specular_exp])
            return mat_ids[k]

        def item(p):
            if isinstance(p, TriangleBatch):
                if id(p) not in batch_ids:
                    batch_ids[id(p)] = len(batch_recs)
                    batch_recs.append([t._record() for t in p._tris])
                    batch_mats.append([mat(t.material) for t in p._tris])
                return (batch_ids[id(p)] << 2) | _lib.KIND_BATCH
            if isinstance(p, Triangle):
                if id(p) not in tri_ids:
                    tri_ids[id(p)] = len(tri_recs)
                    tri_recs.append(p._record())
                    tri_mats.append(mat(p.material))
                return (tri_ids[id(p)] << 2) | _lib.KIND_TRIANGLE
            if id(p) not in solid_ids:
                solid_ids[id(p)] = len(solid_recs)
                solid_recs.append(np.concatenate([p.orientation._m.ravel(), p.inv_orientation._m.ravel(), p.position._v]))
                solid_types.append(p.type)
                solid_mats.append(mat(p.material))
            return (solid_ids[id(p)] << 2) | _lib.KIND_SOLID

        def add(node):
            if node is None:
                return -1
            if node.dimension != n:
                raise TypeError("boundary and data must have the same dimension")
            idx = len(nodes)
            nodes.append(None)
            if isinstance(node, KDLeaf):
                start = len(items)
                for p in node._items:
                    items.append(item(p))
                nodes[idx] = (-1, 0.0, start, len(node._items))
            else:
                l = add(node.left)
                r = add(node.right)
                nodes[idx] = (node.axis, node.split, l, r)
            return idx

        root_idx = add(root)
        nd = np.asarray(nodes, np.float64).reshape(-1, 4)
        rl = n * n + n + 1
        return dict(dimension=n, root=root_idx,
                    node_axis=nd[:, 0].astype(np.int32), node_split=nd[:, 1].astype(f32),
                    node_left=nd[:, 2].astype(np.int32), node_right=nd[:, 3].astype(np.int32),
                    items=np.asarray(items, np.int32),
                    batch_recs=np.asarray(batch_recs, f32).reshape(-1, BATCH_SIZE, rl),
                    batch_mats=np.asarray(batch_mats, np.int32).reshape(-1, BATCH_SIZE),
                    tri_recs=np.asarray(tri_recs, f32).reshape(-1, rl), tri_mats=np.asarray(tri_mats, np.int32),
                    solid_recs=np.asarray(solid_recs, f32).reshape(-1, 2 * n * n + n),
                    solid_types=np.asarray(solid_types, np.int32), solid_mats=np.asarray(solid_mats, np.int32),
                    materials=np.asarray(mats, f32).reshape(-1, 10),
                    aabb_start=boundary.start._v, aabb_end=boundary.end._v)

    def with_rebuilt_tree(self, **kwds):
        """A new CompositeScene over the SAME primitive tables (batches, triangles, solids, materials -- so the same
        pixels: nearest hits do not depend on the tree) with the k-d tree rebuilt by the native builder.  Keyword
        arguments as for build_kdtree (max_depth, split_threshold, traversal_cost, intersection_cost).  Scene
        parameters (lights, background, fov, camera) are copied.  With shadows on the pixels DO depend on the tree,
        here as in the reference: its occlusion walk skips the far child of a branch whenever the split lies nearer
        than the light (tracer.hpp:1298), so a different tree finds different blockers."""
        f = self._flat
        n = self.dimension
        items = []
        brec = np.asarray(f["batch_recs"], np.float64).reshape(-1, BATCH_SIZE, n * n + n + 1)
        if len(brec):
            bv = builder.vertices_of_many(brec.reshape(-1, brec.shape[2]), n).reshape(len(brec), BATCH_SIZE, n, n)
            for k in range(len(brec)):
                items.append(builder._Item((k << 2) | _lib.KIND_BATCH, bv[k].min(axis=(0, 1)), bv[k].max(axis=(0, 1)), bv[k]))
        trec = np.asarray(f["tri_recs"], np.float64).reshape(-1, n * n + n + 1)
        if len(trec):
            tv = builder.vertices_of_many(trec, n)
            for k in range(len(trec)):
                items.append(builder._Item((k << 2) | _lib.KIND_TRIANGLE, tv[k].min(axis=0), tv[k].max(axis=0), tv[k][None]))
        srec = np.asarray(f["solid_recs"], np.float64).reshape(-1, 2 * n * n + n)
        for k in range(len(srec)):
            lo, hi = builder.solid_bounds(int(f["solid_types"][k]) == CUBE, srec[k, 2 * n * n:], srec[k, :n * n].reshape(n, n))
            items.append(builder._Item((k << 2) | _lib.KIND_SOLID, lo, hi))
        nodes, leaf_items = [], []

        def leaf(prims):
            nodes.append([-1, 0.0, len(leaf_items), len(prims)])
            leaf_items.extend(prims)
            return len(nodes) - 1

        def branch(axis, split, left, right):
            nodes.append([axis, split, -1 if left is None else left, -1 if right is None else right])
            return len(nodes) - 1

        lo, hi, root = builder.build_tree(items, leaf, branch, int(kwds.pop("max_depth", builder.KD_DEFAULT_MAX_DEPTH)),
                                          int(kwds.pop("split_threshold", builder.KD_DEFAULT_SPLIT_THRESHOLD)),
                                          float(kwds.pop("traversal_cost", 0.0)), float(kwds.pop("intersection_cost", 0.0)))
        if kwds:
            raise TypeError("unexpected keyword argument %r" % next(iter(kwds)))
        nd = np.asarray(nodes, np.float64).reshape(-1, 4)
        d = {k: np.array(v) for k, v in f.items()}
        d.update(root=root, node_axis=nd[:, 0].astype(np.int32), node_split=nd[:, 1].astype(f32),
                 node_left=nd[:, 2].astype(np.int32), node_right=nd[:, 3].astype(np.int32),
                 items=np.asarray(leaf_items, np.int32), aabb_start=lo, aabb_end=hi)
        other = CompositeScene.from_flat(n, d)
        other.set_fov(self.fov)
        other._push(_pl=list(self._point_lights), _gl=list(self._global_lights), **dict(self._p))
        cam = self.get_camera()
        other._set_camera_arrays(cam._origin, cam._axes)
        return other

    @property
    def root(self):
        """CompositeScene.root (ntracer_body.hpp:922-924): the k-d tree as KDBranch / KDLeaf objects over Triangle,
        TriangleBatch and Solid objects.  Scenes made from flat arrays (build_composite_scene, from_flat) only
        materialise these when asked."""
        if getattr(self, "_root_obj", None) is not None:
            return self._root_obj
        f = self._flat
        n = self._n
        rl = n * n + n + 1
        mats = [Material.__new__(Material) for _ in range(len(f["materials"]))]
        for m, v in zip(mats, np.asarray(f["materials"], np.float64).reshape(-1, 10)):
            m.color = Color(*v[0:3])
            m.specular = Color(*v[3:6])
            m.opacity, m.reflectivity, m.specular_intensity, m.specular_exp = (float(x) for x in v[6:10])

        def tri(rec, mi):
            rec = np.asarray(rec, f32)
            return Triangle(rec[1 + n:1 + 2 * n], rec[1:1 + n], rec[1 + 2 * n:rl].reshape(n - 1, n), mats[int(mi)])

        brecs = np.asarray(f["batch_recs"], f32).reshape(-1, BATCH_SIZE, rl)
        bmats = np.asarray(f["batch_mats"], np.int64).reshape(-1, BATCH_SIZE)
        batches = [TriangleBatch([tri(brecs[k, l], bmats[k, l]) for l in range(BATCH_SIZE)]) for k in range(len(brecs))]
        trecs = np.asarray(f["tri_recs"], f32).reshape(-1, rl)
        tris = [tri(trecs[k], f["tri_mats"][k]) for k in range(len(trecs))]
        srecs = np.asarray(f["solid_recs"], f32).reshape(-1, 2 * n * n + n)
        solids = [Solid(int(f["solid_types"][k]), srecs[k, 2 * n * n:], Matrix._wrap(srecs[k, :n * n].reshape(n, n)), mats[int(f["solid_mats"][k])])
                  for k in range(len(srecs))]
        tables = (batches, tris, solids)
        axis, split, left, right, items = f["node_axis"], f["node_split"], f["node_left"], f["node_right"], f["items"]
        made = {}
        order, stack = [], [int(f["root"])] if int(f["root"]) >= 0 else []
        while stack:
            k = stack.pop()
            order.append(k)
            if axis[k] >= 0:
                stack.extend(int(c) for c in (left[k], right[k]) if c >= 0)
        for k in reversed(order):
            if axis[k] < 0:
                made[k] = KDLeaf([tables[int(it) & 3][int(it) >> 2] for it in items[int(left[k]):int(left[k]) + int(right[k])]])
            else:
                made[k] = KDBranch(int(axis[k]), float(split[k]), made.get(int(left[k])), made.get(int(right[k])))
        self._root_obj = made.get(int(f["root"]))
        return self._root_obj

    def _flat_description(self):
        """The flat arrays this scene was created from (layout of nt_scene_desc / tests/golden/*.npz)."""
        return {k: np.array(v) for k, v in self._flat.items()}

    def _create(self, d):
        self._flat = {k: np.array(d[k]) for k in _FLAT_KEYS}
        n = int(d["dimension"])
        rl = n * n + n + 1
        keep = {}

        def fa(k, shape=None):
            a = np.ascontiguousarray(d[k], f32)
            if shape is not None:
                a = a.reshape(shape)
            keep[k] = a
            return a

        def ia(k):
            a = np.ascontiguousarray(d[k], np.int32)
            keep[k] = a
            return a

        desc = _lib.NtSceneDesc()
        desc.dimension = n
        desc.root = int(d["root"])
        na = ia("node_axis")
        desc.n_nodes = len(na)
        desc.node_axis = na.ctypes.data_as(_lib.i32p)
        desc.node_split = fa("node_split").ctypes.data_as(_lib.f32p)
        desc.node_left = ia("node_left").ctypes.data_as(_lib.i32p)
        desc.node_right = ia("node_right").ctypes.data_as(_lib.i32p)
        it = ia("items")
        desc.n_items = len(it)
        desc.items = it.ctypes.data_as(_lib.i32p)
        br = fa("batch_recs", (-1, BATCH_SIZE, rl))
        desc.n_batches = br.shape[0]
        desc.batch_recs = br.ctypes.data_as(_lib.f32p)
        bm = ia("batch_mats")
        if bm.size != br.shape[0] * BATCH_SIZE:
            raise ValueError("batch_mats does not match batch_recs")
        desc.batch_mats = bm.ctypes.data_as(_lib.i32p)
        tr = fa("tri_recs", (-1, rl))
        desc.n_triangles = tr.shape[0]
        desc.tri_recs = tr.ctypes.data_as(_lib.f32p)
        tm = ia("tri_mats")
        if tm.size != tr.shape[0]:
            raise ValueError("tri_mats does not match tri_recs")
        desc.tri_mats = tm.ctypes.data_as(_lib.i32p)
        sr = fa("solid_recs", (-1, 2 * n * n + n))
        desc.n_solids = sr.shape[0]
        desc.solid_recs = sr.ctypes.data_as(_lib.f32p)
        st = ia("solid_types")
        sm = ia("solid_mats")
        if st.size != sr.shape[0] or sm.size != sr.shape[0]:
            raise ValueError("solid arrays do not match")
        desc.solid_types = st.ctypes.data_as(_lib.i32p)
        desc.solid_mats = sm.ctypes.data_as(_lib.i32p)
        mt = fa("materials", (-1, 10))
        marr = (_lib.NtMaterial * max(len(mt), 1))()
        for i, r in enumerate(mt):
            marr[i].color[:] = [float(v) for v in r[0:3]]
            marr[i].specular[:] = [float(v) for v in r[3:6]]
            marr[i].opacity, marr[i].reflectivity, marr[i].specular_intensity, marr[i].specular_exp = [float(v) for v in r[6:10]]
        desc.n_materials = len(mt)
        desc.materials = marr
        a0 = fa("aabb_start", (n,))
        a1 = fa("aabb_end", (n,))
        desc.aabb_start = a0.ctypes.data_as(_lib.f32p)
        desc.aabb_end = a1.ctypes.data_as(_lib.f32p)
        h = _lib.lib().nt_composite_scene_create(C.byref(desc))
        if not h:
            raise ValueError(_lib.last_error())
        self._handle = h
        self._init_common(n)
        self.boundary = AABB(n, a0, a1)
        # composite_scene defaults (tracer.hpp:1727-1740)
        self._p = dict(shadows=False, camera_light=True, max_reflect_depth=4, bg_gradient_axis=1,
                       ambient=Color(0, 0, 0), bg1=Color(1, 1, 1), bg2=Color(0, 0, 0), bg3=Color(0, 1, 1))
        self._point_lights = []
        self._global_lights = []

    # ---- attribute surface of ntracer_body.hpp:833-933 ----
    shadows = property(lambda s: s._p["shadows"])
    camera_light = property(lambda s: s._p["camera_light"])
    max_reflect_depth = property(lambda s: s._p["max_reflect_depth"])
    bg_gradient_axis = property(lambda s: s._p["bg_gradient_axis"])
    ambient_color = property(lambda s: s._p["ambient"])
    bg1 = property(lambda s: s._p["bg1"])
    bg2 = property(lambda s: s._p["bg2"])
    bg3 = property(lambda s: s._p["bg3"])
    point_lights = property(lambda s: tuple(s._point_lights))
    global_lights = property(lambda s: tuple(s._global_lights))

    def _push(self, **changes):
        p = dict(self._p)
        p.update(changes)
        pls = changes.get("_pl", self._point_lights)
        gls = changes.get("_gl", self._global_lights)
        n = self._n
        sp = _lib.NtSceneParams()
        sp.shadows = 1 if p["shadows"] else 0
        sp.camera_light = 1 if p["camera_light"] else 0
        sp.max_reflect_depth = int(p["max_reflect_depth"])
        sp.bg_gradient_axis = int(p["bg_gradient_axis"])
        sp.ambient[:] = tuple(p["ambient"])
        sp.bg1[:] = tuple(p["bg1"])
        sp.bg2[:] = tuple(p["bg2"])
        sp.bg3[:] = tuple(p["bg3"])
        pp = np.asarray([list(l.position) for l in pls], f32).reshape(-1, n)
        pc = np.asarray([list(l.color) for l in pls], f32).reshape(-1, 3)
        gd = np.asarray([list(l.direction) for l in gls], f32).reshape(-1, n)
        gc = np.asarray([list(l.color) for l in gls], f32).reshape(-1, 3)
        sp.n_point_lights = len(pls)
        sp.point_light_pos = pp.ctypes.data_as(_lib.f32p)
        sp.point_light_color = pc.ctypes.data_as(_lib.f32p)
        sp.n_global_lights = len(gls)
        sp.global_light_dir = gd.ctypes.data_as(_lib.f32p)
        sp.global_light_color = gc.ctypes.data_as(_lib.f32p)
        _lib.check(_lib.lib().nt_scene_set_params(self._handle, C.byref(sp)))
        for k, v in changes.items():
            if not k.startswith("_"):
                self._p[k] = v
        self._point_lights = list(pls)
        self._global_lights = list(gls)

    def set_shadows(self, v):
        self._push(shadows=bool(v))

    def set_camera_light(self, v):
        self._push(camera_light=bool(v))

    def set_max_reflect_depth(self, v):
        self._push(max_reflect_depth=int(v))

    def set_ambient_color(self, color):
        self._push(ambient=Color._coerce(color))

    def set_background(self, c1, c2=None, c3=None, axis=1):
        c1 = Color._coerce(c1)
        c2 = c1 if c2 is None else Color._coerce(c2)
        c3 = c1 if c3 is None else Color._coerce(c3)
        axis = int(axis)
        if axis < 0 or axis >= self._n:
            raise ValueError('"axis" must be between 0 and one less than the dimension of the scene')
        self._push(bg1=c1, bg2=c2, bg3=c3, bg_gradient_axis=axis)

    def add_light(self, light):
        if isinstance(light, PointLight):
            if light.dimension != self._n:
                raise TypeError("the light must have the same dimension as the scene")
            self._push(_pl=self._point_lights + [light])
        elif isinstance(light, GlobalLight):
            if light.dimension != self._n:
                raise TypeError("the light must have the same dimension as the scene")
            self._push(_gl=self._global_lights + [light])
        else:
            raise TypeError("object must be an instance of PointLight or GlobalLight")

    def set_params_flat(self, p):
        """Apply a parameter dictionary in the layout of the fixtures (tools/gen_golden.py scene_params)."""
        n = self._n
        pls = [PointLight(Vector(n, pos), tuple(col)) for pos, col in
               zip(np.asarray(p["point_light_pos"], f32).reshape(-1, n), np.asarray(p["point_light_color"], f32).reshape(-1, 3))]
        gls = [GlobalLight(Vector(n, d), tuple(col)) for d, col in
               zip(np.asarray(p["global_light_dir"], f32).reshape(-1, n), np.asarray(p["global_light_color"], f32).reshape(-1, 3))]
        if "fov" in p:
            self.set_fov(float(p["fov"]))
        self._push(shadows=bool(p["shadows"]), camera_light=bool(p["camera_light"]),
                   max_reflect_depth=int(p["max_reflect_depth"]), bg_gradient_axis=int(p["bg_gradient_axis"]),
                   ambient=Color._coerce(tuple(float(v) for v in p["ambient"])),
                   bg1=Color._coerce(tuple(float(v) for v in p["bg1"])), bg2=Color._coerce(tuple(float(v) for v in p["bg2"])),
                   bg3=Color._coerce(tuple(float(v) for v in p["bg3"])), _pl=pls, _gl=gls)


def screen_coord_to_ray(cam, x, y, w, h, fov):
    """tracern.screen_coord_to_ray(cam,x,y,w,h,fov): flat_origin_ray_source (tracer.hpp:60-76), fp32."""
    n = cam.dimension
    half_w = f32(w) / f32(2)
    half_h = f32(h) / f32(2)
    fovI = f32(math.tan(f32(fov) / f32(2))) / half_w
    sx = f32(fovI * f32(f32(x) - half_w))
    sy = f32(fovI * f32(f32(y) - half_h))
    v = (cam._axes[2] + cam._axes[0] * sx) - cam._axes[1] * sy
    return Vector._wrap(v).unit()


def cross(vectors):
    """tracern.cross(vectors) -- generalised cross product (geometry.hpp:884-892)."""
    vs = list(vectors)
    return Vector._wrap(builder.cross([v._v if isinstance(v, Vector) else list(v) for v in vs]).astype(f32))


def _clip_overlaps(n, verts, tight, shared, first_free, lo, hi):
    """nt_polytope_clip_box against the box shrunk by a hair: what merely touches the box is left with nothing."""
    v = np.ascontiguousarray(verts, f32)
    t = np.ascontiguousarray(tight, np.uint64)
    scale = max(1.0, float(np.abs(v).max()), float(np.abs(lo).max()), float(np.abs(hi).max()))
    eps = 1e-6 * scale
    l = np.ascontiguousarray(lo + eps, f32)
    h = np.ascontiguousarray(hi - eps, f32)
    if (l >= h).any():
        return False
    r = _lib.check(_lib.lib().nt_polytope_clip_box(n, len(v), v.ctypes.data_as(_lib.f32p), t.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   shared, first_free, l.ctypes.data_as(_lib.f32p), h.ctypes.data_as(_lib.f32p), None, None))
    return r > 0


def _simplex_tight(n):
    t = np.zeros((n, 3), np.uint64)
    for i in range(n):
        for j in range(n):
            if j != i:
                t[i, j >> 6] |= np.uint64(1) << np.uint64(j & 63)
    return t


class PrimitivePrototype(object):
    """Base of the objects build_kdtree consumes: a primitive plus its bounding box (tracer.hpp:1363-1373)."""
    boundary = None
    primitive = None

    @property
    def dimension(self):
        return self.boundary.dimension

    @property
    def material(self):
        return self.primitive.material

    def _overlaps(self, lo, hi):
        return bool(((self.boundary.start._v < hi) & (self.boundary.end._v > lo)).all())


class TrianglePointData(object):
    """One vertex of a TrianglePrototype: ``point`` and the edge normal that belongs to it (tracer.hpp:1384-1389)."""
    __slots__ = ("point", "edge_normal")

    def __init__(self, point, edge_normal):
        self.point = point
        self.edge_normal = edge_normal

    def __iter__(self):                    # lets callers that expect plain points unpack the coordinates
        return iter(self.point)


def _point_data(tri, pts):
    # vertex 0 belongs to the edge normal that is minus the sum of the others (the barycentric gradients add up to 0)
    first = Vector._wrap(-np.sum([e._v for e in tri.edge_normals], axis=0))
    return tuple(TrianglePointData(Vector._wrap(pts[k]), first if k == 0 else tri.edge_normals[k - 1]) for k in range(len(pts)))


class TrianglePrototype(PrimitivePrototype):
    """tracern.TrianglePrototype(points[,material]) or TrianglePrototype(triangle) -- tracer.hpp:1391-1405,
    ntracer_body.hpp:2680-2760."""

    def __init__(self, points, material=None):
        if isinstance(points, Triangle):
            if material is not None:
                raise TypeError("material is taken from the triangle")
            self.primitive = points
            a = builder.vertices_of(points.p1._v, points.face_normal._v, [e._v for e in points.edge_normals])
        else:
            pts = [list(p) for p in points]
            if material is None:
                raise TypeError("material is required")
            self.primitive = Triangle.from_points(pts, material)
            a = np.asarray(pts, f32)
        n = a.shape[1]
        self.boundary = AABB(n, a.min(axis=0), a.max(axis=0))
        self.face_normal = self.primitive.face_normal
        self.point_data = _point_data(self.primitive, a)

    def _vertices(self):
        return np.asarray([list(pd.point) for pd in self.point_data], f32)

    def _overlaps(self, lo, hi):
        n = self.dimension
        if not PrimitivePrototype._overlaps(self, lo, hi):
            return False
        return _clip_overlaps(n, self._vertices(), _simplex_tight(n), n - 2, n, lo, hi)


class _LaneView(object):
    """x[i] of a batch prototype's per-lane attribute"""
    __slots__ = ("_items",)

    def __init__(self, items):
        self._items = tuple(items)

    def __getitem__(self, i):
        return self._items[i]

    def __len__(self):
        return len(self._items)


class TriangleBatchPointData(object):
    __slots__ = ("point", "edge_normal")

    def __init__(self, point, edge_normal):
        self.point = point
        self.edge_normal = edge_normal


class TriangleBatchPrototype(PrimitivePrototype):
    """tracern.TriangleBatchPrototype(triangle_prototypes | TriangleBatch) -- tracer.hpp:1406-1437: BATCH_SIZE
    triangle prototypes side by side; ``face_normal[i]``, ``point_data[j].point[i]``, ``point_data[j].edge_normal[i]``
    and ``material[i]`` address lane i."""

    def __init__(self, t_prototypes):
        if isinstance(t_prototypes, TriangleBatch):
            protos = [TrianglePrototype(t) for t in t_prototypes]
            self.primitive = t_prototypes
        else:
            protos = list(t_prototypes)
            if len(protos) != BATCH_SIZE or not all(isinstance(p, TrianglePrototype) for p in protos):
                raise ValueError("exactly %d TrianglePrototype instances are required" % BATCH_SIZE)
            self.primitive = TriangleBatch([p.primitive for p in protos])
        n = protos[0].dimension
        if any(p.dimension != n for p in protos):
            raise TypeError("the prototypes must have the same dimension")
        self._protos = tuple(protos)
        self.boundary = AABB(n, np.min([p.boundary.start._v for p in protos], axis=0), np.max([p.boundary.end._v for p in protos], axis=0))
        self.face_normal = _LaneView(p.face_normal for p in protos)
        self.point_data = tuple(TriangleBatchPointData(_LaneView(p.point_data[j].point for p in protos),
                                                       _LaneView(p.point_data[j].edge_normal for p in protos)) for j in range(n))

    @property
    def material(self):
        return _LaneView(p.material for p in self._protos)

    def _overlaps(self, lo, hi):
        return PrimitivePrototype._overlaps(self, lo, hi) and any(p._overlaps(lo, hi) for p in self._protos)


def triangle_prototypes(simplices, material):
    """Many TrianglePrototypes at once from a (count, n, n) array of vertices (not in the reference: its callers
    loop over TrianglePrototype(points, material); this is that loop with the records computed in bulk)."""
    if not isinstance(material, Material):
        raise TypeError("material must be a Material")
    a = np.ascontiguousarray(simplices, f32)
    p1, fn, edges = builder.from_points_records(a)
    n = a.shape[2]
    out = []
    for k in range(len(a)):
        tp = object.__new__(TrianglePrototype)
        tp.primitive = Triangle(p1[k], fn[k], edges[k], material)
        tp.boundary = AABB(n, a[k].min(axis=0), a[k].max(axis=0))
        tp.face_normal = tp.primitive.face_normal
        tp.point_data = _point_data(tp.primitive, a[k])
        out.append(tp)
    return out


class SolidPrototype(PrimitivePrototype):
    """tracern.SolidPrototype(type,position,orientation,material) -- tracer.hpp:1375-1382."""

    def __init__(self, type, position, orientation, material):
        self.primitive = Solid(type, position, orientation, material)
        lo, hi = builder.solid_bounds(type == CUBE, self.primitive.position._v, orientation._m)
        self.boundary = AABB(orientation.dimension, lo, hi)
        self.type = type
        self.position = self.primitive.position
        self.orientation = orientation
        self.inv_orientation = self.primitive.inv_orientation

    def _overlaps(self, lo, hi):
        n = self.dimension
        if not PrimitivePrototype._overlaps(self, lo, hi):
            return False
        o = self.orientation._m.astype(np.float64)
        pos = self.position._v.astype(np.float64)
        if self.type == CUBE:
            # x = O (u + p), u in [-1,1]^n: a parallelotope; vertex facets: bit 2k (+1 for the upper side of axis k)
            signs = np.array([[1.0 if (m >> k) & 1 else -1.0 for k in range(n)] for m in range(1 << n)])
            verts = (signs + pos) @ o.T
            tight = np.zeros((1 << n, 3), np.uint64)
            for m in range(1 << n):
                for k in range(n):
                    bit = 2 * k + ((m >> k) & 1)
                    tight[m, bit >> 6] |= np.uint64(1) << np.uint64(bit & 63)
            return _clip_overlaps(n, verts, tight, n - 1, 2 * n, lo, hi)
        # sphere: minimise |O^-1 x - p|^2 over the box (convex) by projected gradient from the clamped centre
        inv = self.inv_orientation._m.astype(np.float64)
        x = np.clip(o @ pos, lo, hi)
        step = 1.0 / max(1e-12, 2.0 * np.linalg.norm(inv, 2) ** 2)
        for _ in range(200):
            g = 2.0 * inv.T @ (inv @ x - pos)
            x = np.clip(x - step * g, lo, hi)
        return bool(np.linalg.norm(inv @ x - pos) < 1.0 - 1e-9)


def build_kdtree(primitives, extra_threads=-1, **kwds):
    """tracern.build_kdtree(primitives[,extra_threads=-1,*,update_primitives=False]) -> (AABB, KDNode)
    (ntracer_body.hpp:3250-3325, tracer.hpp:2431-2455).  ``extra_threads`` is accepted for compatibility."""
    max_depth = int(kwds.pop("max_depth", builder.KD_DEFAULT_MAX_DEPTH))
    split_threshold = int(kwds.pop("split_threshold", builder.KD_DEFAULT_SPLIT_THRESHOLD))
    kwds.pop("update_primitives", None)
    kwds_flat = bool(kwds.pop("_flat", False))
    traversal_cost = float(kwds.pop("traversal_cost", 0.0) or 0.0)
    intersection_cost = float(kwds.pop("intersection_cost", 0.0) or 0.0)
    if kwds:
        raise TypeError("unexpected keyword argument %r" % next(iter(kwds)))
    protos = list(primitives)
    if not protos:
        raise ValueError("cannot build tree from empty sequence")
    for p in protos:
        if not isinstance(p, PrimitivePrototype):
            raise TypeError("object is not an instance of PrimitivePrototype")
    n = protos[0].dimension
    if any(p.dimension != n for p in protos):
        raise TypeError("the primitive prototypes must all have the same dimension")
    tri = [builder._Item(p.primitive, p.boundary.start._v, p.boundary.end._v, [p._vertices()]) for p in protos
           if isinstance(p, TrianglePrototype)]
    other = [builder._Item(p.primitive, p.boundary.start._v, p.boundary.end._v) for p in protos if not isinstance(p, TrianglePrototype)]
    batches, loose = builder.group_batches(tri, BATCH_SIZE, TriangleBatch)
    items = batches + loose + other
    if kwds_flat:
        return n, items, builder.build_tree_arrays(items, max_depth, split_threshold, traversal_cost, intersection_cost)
    lo, hi, root = builder.build_tree(items, KDLeaf, KDBranch, max_depth, split_threshold, traversal_cost, intersection_cost)
    return AABB(n, lo, hi), root


def build_composite_scene(primitives, extra_threads=-1, **kwds):
    """tracern.build_composite_scene(primitives[,extra_threads=-1,*,update_primitives=False]) -> CompositeScene
    (ntracer_body.hpp:3335-3357).  The tree goes from the native builder's arrays straight into the scene
    description; KDBranch / KDLeaf objects are not made on the way."""
    n, items, (lo, hi, axis, split, left, right, leaf_items, root) = build_kdtree(primitives, extra_threads, _flat=True, **kwds)
    # primitive tables in item order, materials interned by value
    mats, mat_ids = [], {}

    def mat(m):
        k = m._key()
        if k not in mat_ids:
            mat_ids[k] = len(mats)
            mats.append(list(m.color) + list(m.specular) + [m.opacity, m.reflectivity, m.specular_intensity, m.specular_exp])
        return mat_ids[k]

    rl = n * n + n + 1
    codes = np.zeros(len(items), np.int32)
    batch_recs, batch_mats, tri_recs, tri_mats, solid_recs, solid_types, solid_mats = [], [], [], [], [], [], []
    for k, it in enumerate(items):
        p = it.prim
        if isinstance(p, TriangleBatch):
            codes[k] = (len(batch_recs) << 2) | _lib.KIND_BATCH
            batch_recs.append([t._record() for t in p._tris])
            batch_mats.append([mat(t.material) for t in p._tris])
        elif isinstance(p, Triangle):
            codes[k] = (len(tri_recs) << 2) | _lib.KIND_TRIANGLE
            tri_recs.append(p._record())
            tri_mats.append(mat(p.material))
        else:
            codes[k] = (len(solid_recs) << 2) | _lib.KIND_SOLID
            solid_recs.append(np.concatenate([p.orientation._m.ravel(), p.inv_orientation._m.ravel(), p.position._v]))
            solid_types.append(p.type)
            solid_mats.append(mat(p.material))
    d = dict(dimension=n, root=int(root), node_axis=np.asarray(axis, np.int32), node_split=np.asarray(split, f32),
             node_left=np.asarray(left, np.int32), node_right=np.asarray(right, np.int32),
             items=codes[np.asarray(leaf_items, np.int64)] if len(leaf_items) else np.zeros(0, np.int32),
             batch_recs=np.asarray(batch_recs, f32).reshape(-1, BATCH_SIZE, rl), batch_mats=np.asarray(batch_mats, np.int32).reshape(-1, BATCH_SIZE),
             tri_recs=np.asarray(tri_recs, f32).reshape(-1, rl), tri_mats=np.asarray(tri_mats, np.int32),
             solid_recs=np.asarray(solid_recs, f32).reshape(-1, 2 * n * n + n), solid_types=np.asarray(solid_types, np.int32),
             solid_mats=np.asarray(solid_mats, np.int32), materials=np.asarray(mats, f32).reshape(-1, 10),
             aabb_start=np.asarray(lo, f32), aabb_end=np.asarray(hi, f32))
    return CompositeScene.from_flat(n, d)

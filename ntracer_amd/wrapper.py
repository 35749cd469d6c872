"""Mirror of the reference's ``ntracer.wrapper`` (lib/ntracer/wrapper.py:71-147): ``NTracer(dimension)``
pre-binds ``dimension`` on every constructor that takes it."""
import weakref

from . import tracern
from .tracern import CUBE, SPHERE  # noqa: F401


def _bind_dimension(base, dim, varargs=False):
    class Bound(base):
        __doc__ = base.__doc__

        def __init__(self, *args, **kwds):
            if varargs and len(args) > 1:
                args = (args,)
            base.__init__(self, dim, *args, **kwds)

    Bound.__name__ = base.__name__
    return Bound


class _VectorFactory(object):
    """nt.Vector(...) / nt.Vector.axis(axis,length) with the dimension filled in."""

    def __init__(self, dim):
        self._dim = dim

    def __call__(self, *values):
        if len(values) == 1 and not isinstance(values[0], (int, float)):
            values = list(values[0])           # any iterable, generators included
        if len(values) == 0:
            return tracern.Vector(self._dim)
        return tracern.Vector(self._dim, values)

    def axis(self, axis, length=1):
        return tracern.Vector.axis(self._dim, axis, length)


class _MatrixFactory(object):
    def __init__(self, dim):
        self._dim = dim

    def __call__(self, *values):
        if len(values) == 1:
            values = list(values[0])
        return tracern.Matrix(self._dim, values)

    def identity(self):
        return tracern.Matrix.identity(self._dim)

    def scale(self, *a):
        return tracern.Matrix.scale(self._dim, a[0] if len(a) == 1 else a)

    def rotation(self, a, b, theta):
        return tracern.Matrix.rotation(a, b, theta)

    def reflection(self, a):
        return tracern.Matrix.reflection(a)


class NTracer(object):
    """NTracer(dimension[,force_generic=False]): helper that creates objects of one dimension.
    ``force_generic`` is accepted for compatibility: the fixed-N vs run-time-n choice is made inside
    the HIP library (template<int N> kernels for 3..10, LDS-staged run-time-n kernel otherwise)."""
    _cache = weakref.WeakValueDictionary()

    def __new__(cls, dimension, force_generic=False):
        obj = None if force_generic else NTracer._cache.get(dimension)
        if obj is not None:
            return obj
        obj = object.__new__(cls)
        obj.dimension = dimension
        obj.base = tracern
        obj.Vector = _VectorFactory(dimension)
        obj.Matrix = _MatrixFactory(dimension)
        obj.Camera = _bind_dimension(tracern.Camera, dimension)
        obj.BoxScene = _bind_dimension(tracern.BoxScene, dimension)
        obj.AABB = _bind_dimension(tracern.AABB, dimension)
        for n in ("CompositeScene", "KDNode", "KDLeaf", "KDBranch", "Primitive", "PrimitiveBatch", "Solid", "Triangle",
                  "TriangleBatch", "TrianglePrototype", "TriangleBatchPrototype", "SolidPrototype", "PrimitivePrototype", "PointLight",
                  "GlobalLight",
                  "dot", "cross", "build_kdtree", "build_composite_scene",
                  "screen_coord_to_ray", "BATCH_SIZE"):
            setattr(obj, n, getattr(tracern, n))
        if not force_generic:
            NTracer._cache[dimension] = obj
        return obj

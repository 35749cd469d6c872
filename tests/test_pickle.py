"""Pickling in the reference's wire format (src/render.cpp:1391-1477, 1482-1660,
1696-1751): the byte strings in tests/golden/pickles.npz were written by the compiled reference
(tools/gen_golden.py --only pickles).  With the module aliases installed they must load here to the recorded
values, and our own pickles must be byte-identical to the reference's (same reduce tuples, protocol 2)."""
import pickle

import numpy as np
import pytest

import fixtures as fx
import ntracer_amd
from ntracer_amd import compat, render, tracern


@pytest.fixture()
def aliases():
    compat.alias_reference_modules()
    yield
    compat.remove_aliases()


def _values(obj):
    if isinstance(obj, render.Color):
        return list(obj)
    if isinstance(obj, render.Material):
        return list(obj.color) + list(obj.specular) + [obj.opacity, obj.reflectivity, obj.specular_intensity, obj.specular_exp]
    if isinstance(obj, tracern.Vector):
        return list(obj)
    if isinstance(obj, tracern.Matrix):
        return obj._m
    if isinstance(obj, tracern.AABB):
        return [list(obj.start), list(obj.end)]
    if isinstance(obj, tracern.Triangle):
        return obj._rows()
    if isinstance(obj, tracern.TriangleBatch):
        return [t._rows() for t in obj]
    if isinstance(obj, tracern.Solid):
        return list(obj.orientation._m.ravel()) + list(obj.position)
    raise TypeError(type(obj))


def test_reference_pickles_load_and_round_trip_byte_for_byte(aliases):
    g = fx.load("pickles")
    for name in g["names"]:
        name = str(name)
        blob = bytes(g["pickle_" + name])
        obj = pickle.loads(blob)
        want = g["values_" + name]
        got = np.asarray(_values(obj), np.float32).reshape(want.shape)
        assert np.array_equal(got, want), name
        again = pickle.dumps(obj, 2).replace(b"ntracer_amd.render", b"ntracer.render")
        assert again == blob, name
    so = pickle.loads(bytes(g["pickle_solid5"]))
    assert so.type == ntracer_amd.SPHERE and so.material.specular_exp == 12.0
    tb = pickle.loads(bytes(g["pickle_batch9"]))
    assert tb[0].material is tb[3].material            # the memoised Material is shared, as in the reference


def test_own_pickles_without_aliases():
    nt = ntracer_amd.NTracer(4)
    m = ntracer_amd.Material((1, .5, .25), .5, .25)
    t = nt.Triangle.from_points(np.eye(4) + .1, m)
    for obj in (render.Color(.1, .2, .3), m, nt.Vector(1, 2, 3, 4), nt.Matrix.identity(), nt.AABB(), t,
                nt.TriangleBatch([t] * 4), nt.Solid(ntracer_amd.CUBE, nt.Vector(0, 0, 0, 1), nt.Matrix.scale(2), m)):
        for proto in (2, pickle.HIGHEST_PROTOCOL):
            back = pickle.loads(pickle.dumps(obj, proto))
            assert isinstance(obj, type(back))          # wrapper subclasses come back as the base type, as in the reference
            assert np.array_equal(np.asarray(_values(back), np.float32).ravel(), np.asarray(_values(obj), np.float32).ravel())
    assert pickle.loads(pickle.dumps(t)).d == t.d


def test_malformed_pickle_payloads_raise_like_the_reference():
    m = ntracer_amd.Material((1, 1, 1))
    with pytest.raises(ValueError, match="color data is malformed"):
        render._color_unpickle(b"\0" * 11)
    with pytest.raises(ValueError, match="material data is malformed"):
        render._material_unpickle(b"\0" * 44)
    with pytest.raises(ValueError, match="dimension cannot be less than 3"):
        render._vector_unpickle(2, b"\0" * 8)
    with pytest.raises(ValueError, match="vector data is malformed"):
        render._vector_unpickle(3, b"\0" * 8)
    with pytest.raises(ValueError, match="matrix data is malformed"):
        render._matrix_unpickle(3, b"\0" * 8)
    with pytest.raises(ValueError, match="triangle data is malformed"):
        render._triangle_unpickle(3, b"\0" * 8, m)
    with pytest.raises(TypeError, match="different batch size"):
        render._triangle_batch_unpickle(8, 3, b"", m, m, m, m)
    with pytest.raises(TypeError, match="wrong number of arguments"):
        render._triangle_batch_unpickle(4, 3, b"", m, m)
    with pytest.raises(ValueError, match="solid data is malformed"):
        render._solid_unpickle(3, b"\1", m)
    with pytest.raises(ValueError, match="solid data is corrupt"):
        render._solid_unpickle(3, b"\7" + b"\0" * 48, m)
    with pytest.raises(ValueError, match="AABB data is malformed"):
        render._aabb_unpickle(3, b"\0")


def test_alias_refuses_to_shadow_a_real_package():
    import sys
    import types
    sys.modules["ntracer"] = types.ModuleType("ntracer")
    try:
        with pytest.raises(RuntimeError):
            compat.alias_reference_modules()
    finally:
        del sys.modules["ntracer"]

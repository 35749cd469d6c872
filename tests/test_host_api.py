"""Host-side mirror of ntracer.tracern / ntracer.render / ntracer.wrapper (no GPU needed)."""
import math

import numpy as np
import pytest

import fixtures as fx
import ntracer_amd
from ntracer_amd import tracern
from ntracer_amd.wrapper import NTracer


def rotation_cameras(nt, cam_distance, frames):
    """The RotatingCamera loop of the reference's scripts/polytope.py:522-556 on OUR Camera/Matrix."""
    n = nt.dimension
    cam = nt.Camera()
    cam.translate(nt.Vector.axis(2, cam_distance) + nt.Vector(*((0, 0, 0) + (0.0001,) * (n - 3))))
    incr = 2 * math.pi / 160
    h = 1 / math.sqrt(n - 1)
    out = []
    for f in range(frames):
        out.append((np.array(list(cam.origin), np.float32), np.array([list(cam.axes[i]) for i in range(n)], np.float32)))
        a2 = cam.axes[0] * h + cam.axes[1] * h
        for i in range(n - 3):
            a2 = a2 + cam.axes[i + 3] * h
        cam.transform(nt.Matrix.rotation(cam.axes[2], a2, incr))
        cam.normalize()
        cam.origin = cam.axes[2] * cam_distance
    return out


@pytest.mark.parametrize("name,n", [("box_n3_1920x1080", 3), ("box_n6_1920x1080", 6), ("box_n10_4096x4096", 10)])
def test_camera_math_reproduces_reference_rotation(name, n):
    g = fx.load(name)
    cams = rotation_cameras(NTracer(n), -math.sqrt(n) * 4, 40)
    for f, (o, a) in enumerate(cams):
        assert np.abs(o - g["origins"][f]).max() < 2e-5, f
        assert np.abs(a - g["axes"][f]).max() < 2e-6, f


def test_vector_matrix_basics():
    nt = NTracer(4)
    v = nt.Vector(1, 2, 3, 4)
    assert list(v + v) == [2, 4, 6, 8] and list(-v) == [-1, -2, -3, -4] and list(v * 2) == [2, 4, 6, 8]
    assert tracern.dot(v, v) == 30 and v.square() == 30 and abs(v.absolute() - math.sqrt(30)) < 1e-6
    assert abs(v.unit().absolute() - 1) < 1e-6
    m = nt.Matrix.rotation(nt.Vector.axis(0), nt.Vector.axis(1), 0.3)
    i = m * m.inverse()
    assert np.abs(np.array(i.values).reshape(4, 4) - np.eye(4)).max() < 1e-6
    assert list(nt.Matrix.identity() * v) == list(v)
    with pytest.raises(TypeError):
        tracern.dot(v, nt.Vector.axis(0) if False else tracern.Vector(3))


def test_kd_objects_flatten_like_the_fixture_layout():
    ka = fx.known_answer()
    mat = ntracer_amd.Material((1, 1, 1))
    nt = NTracer(3)
    prims = [nt.Triangle(t["p1"], t["face_normal"], t["edge_normals"], mat) for t in ka["triangles"]]

    def build(node):
        if node is None:
            return None
        if "leaf" in node:
            return nt.KDLeaf([prims[i] for i in node["leaf"]])
        b = node["branch"]
        return nt.KDBranch(b["axis"], b["split"], build(b["left"]), build(b["right"]))

    scene = nt.CompositeScene(nt.AABB(ka["aabb"]["start"], ka["aabb"]["end"]), build(ka["tree"]))
    scene.set_fov(ka["fov"])
    assert scene.dimension == 3
    flat = tracern.CompositeScene._flatten(nt.AABB(ka["aabb"]["start"], ka["aabb"]["end"]), build(ka["tree"]))
    ref = fx.known_answer_flat(ka)
    assert flat["root"] == ref["root"]
    for k in ("node_axis", "node_left", "node_right"):
        assert np.array_equal(flat[k], ref[k]), k
    assert np.allclose(flat["node_split"], ref["node_split"])
    # leaf items refer to the same triangles in the same order (indices are assigned in visit order)
    order = [prims.index(p) for p in [prims[4], prims[5], prims[2], prims[3], prims[1], prims[0]]]
    recs = flat["tri_recs"]
    for slot, pi in enumerate(order):
        assert np.allclose(recs[slot], ref["tri_recs"][pi])
    with pytest.raises(ValueError):
        nt.KDBranch(0, 1.0)
    with pytest.raises(ValueError):
        nt.KDLeaf([])


def test_triangle_batch_and_materials():
    nt = NTracer(3)
    mat = ntracer_amd.Material((1, .5, .5), 1, .25, .5, 4, (1, 1, 0))
    assert tuple(mat.color) == (1, .5, .5) and mat.reflectivity == .25 and mat.specular_exp == 4
    with pytest.raises(ValueError):
        ntracer_amd.Material((1, 1, 1), opacity=1.5)
    t = nt.Triangle((0, 0, 0), (0, 0, 1), [(1, 0, 0), (0, 1, 0)], mat)
    assert t.d == 0 and t.dimension == 3
    assert nt.BATCH_SIZE == 4
    b = nt.TriangleBatch([t] * 4)
    assert len(b) == 4 and b[2] is t
    with pytest.raises(ValueError):
        nt.TriangleBatch([t] * 3)


def test_composite_scene_attribute_surface():
    g = fx.load("cell600_n4")
    sc = tracern.CompositeScene.from_flat(4, fx.flat_of(g))
    sc.set_shadows(True)
    sc.set_camera_light(False)
    sc.set_max_reflect_depth(2)
    sc.set_ambient_color((.1, .2, .3))
    sc.set_background((1, 0, 0), (0, 1, 0), (0, 0, 1), 2)
    sc.add_light(tracern.PointLight(tracern.Vector(4, (1, 2, 3, 4)), (1, 1, 1)))
    sc.add_light(tracern.GlobalLight(tracern.Vector(4, (0, -1, 0, 0)), (.5, .5, .5)))
    assert sc.shadows and not sc.camera_light and sc.max_reflect_depth == 2 and sc.bg_gradient_axis == 2
    assert tuple(sc.bg3) == (0, 0, 1) and len(sc.point_lights) == 1 and len(sc.global_lights) == 1
    with pytest.raises(ValueError):
        sc.set_background((1, 1, 1), axis=4)
    with pytest.raises(TypeError):
        sc.add_light(tracern.PointLight(tracern.Vector(3, (1, 2, 3)), (1, 1, 1)))
    with pytest.raises(TypeError):
        sc.add_light(object())


def test_scene_root_and_boundary_round_trip():
    """scene.root / scene.boundary (ntracer_body.hpp:919-924): objects rebuilt from the flat description flatten back
    to the same arrays, whether the scene came from flat arrays or from our builder."""
    g = fx.load("feature3d")
    flat = fx.flat_of(g)
    sc = tracern.CompositeScene.from_flat(3, flat)
    assert np.allclose(list(sc.boundary.start), flat["aabb_start"]) and np.allclose(list(sc.boundary.end), flat["aabb_end"])
    again = tracern.CompositeScene._flatten(sc.boundary, sc.root)
    for k in ("node_axis", "node_split", "batch_recs", "tri_recs", "solid_types"):
        assert np.array_equal(np.asarray(again[k]).ravel(), np.asarray(flat[k]).ravel()), k
    # a Solid object recomputes its inverse orientation, which can differ from the stored one in the last bit
    assert np.allclose(np.asarray(again["solid_recs"]).ravel(), np.asarray(flat["solid_recs"]).ravel(), rtol=1e-6, atol=1e-7)
    assert len(again["items"]) == len(flat["items"])
    nt = NTracer(3)
    mat = ntracer_amd.Material((1, .5, .5))
    sc2 = nt.build_composite_scene([nt.TrianglePrototype([(i, 0, 0), (i + 1, 0, 0), (i, 1, .2 * i)], mat) for i in range(9)])
    leaves = []

    def walk(nd):
        if nd is None:
            return
        if isinstance(nd, tracern.KDLeaf):
            leaves.append(nd)
        else:
            walk(nd.left)
            walk(nd.right)

    walk(sc2.root)
    assert leaves and all(isinstance(p, tracern.TriangleBatch) for lf in leaves for p in lf)


def test_vector_set_c_and_matrix_determinant():
    nt = NTracer(4)
    v = nt.Vector(1, 2, 3, 4)
    w = v.set_c(2, 9.5)
    assert list(w) == [1, 2, 9.5, 4] and list(v) == [1, 2, 3, 4]
    with pytest.raises(IndexError):
        v.set_c(4, 0)
    m = nt.Matrix([[2, 0, 0, 0], [0, 3, 0, 0], [0, 0, 4, 0], [1, 1, 1, 5]])
    assert abs(m.determinant() - 120) < 1e-4
    assert abs(nt.Matrix.rotation(nt.Vector.axis(0), nt.Vector.axis(1), .7).determinant() - 1) < 1e-6


def test_wrapper_cache_and_scene_type_rules():
    assert NTracer(5) is NTracer(5)
    assert NTracer(5, force_generic=True) is not NTracer(5)
    with pytest.raises(TypeError):
        ntracer_amd.Scene()
    c = ntracer_amd.Color(.5, .25, 1) * 2 + (0, .5, 0)
    assert tuple(c) == (1, 1, 2)
    d = tracern.screen_coord_to_ray(tracern.Camera(3), 10, 20, 640, 480, 0.8)
    assert abs(d.absolute() - 1) < 1e-6 and d[2] > 0.85


def test_channels_from_surface_layouts():
    from ntracer_amd.pygame_render import channels_from_surface

    class Surf(object):
        def __init__(self, nbytes, losses, shifts, masks):
            self.n, self.l, self.s, self.m = nbytes, losses, shifts, masks
        get_bytesize = lambda s: s.n
        get_losses = lambda s: s.l
        get_shifts = lambda s: s.s
        get_masks = lambda s: s.m

    # 32-bit XRGB (pygame's usual display format): pad8, R8, G8, B8
    ch = channels_from_surface(Surf(4, (0, 0, 0, 8), (16, 8, 0, 0), (0xFF0000, 0xFF00, 0xFF, 0)))
    assert [(c.bit_size, c.f_r, c.f_g, c.f_b) for c in ch] == [(8, 0, 0, 0), (8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1)]
    # RGB565
    ch = channels_from_surface(Surf(2, (3, 2, 3, 8), (11, 5, 0, 0), (0xF800, 0x07E0, 0x001F, 0)))
    assert [(c.bit_size, c.f_r, c.f_g, c.f_b) for c in ch] == [(5, 1, 0, 0), (6, 0, 1, 0), (5, 0, 0, 1)]
    # 24-bit BGR
    ch = channels_from_surface(Surf(3, (0, 0, 0, 8), (0, 8, 16, 0), (0xFF, 0xFF00, 0xFF0000, 0)))
    assert [(c.bit_size, c.f_r, c.f_b) for c in ch] == [(8, 0, 1), (8, 0, 0), (8, 1, 0)]
    # RGBA8888 with alpha in the low byte
    ch = channels_from_surface(Surf(4, (0, 0, 0, 0), (24, 16, 8, 0), (0xFF000000, 0xFF0000, 0xFF00, 0xFF)))
    assert [(c.bit_size, c.f_c) for c in ch] == [(8, 0), (8, 0), (8, 0), (8, 1)]
    with pytest.raises(TypeError):
        channels_from_surface(Surf(1, (0, 0, 0, 0), (0, 0, 0, 0), (0, 0, 0, 0)))

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the compiled reference.

THIS SCRIPT RUNS ONLY IN THE BUILD CONTAINER.  It imports the reference
(`ntracer`, built in a scratch directory outside the repo -- recipe in
DESIGN.md / SURVEY.md section 8c) and records DATA ONLY: cameras, flattened
scenes (k-d nodes, leaf item lists, simplex records, materials), golden fp32
colours from ``Scene.calculate_color`` and packed bytes from
``BlockingRenderer.render``.  Nothing of the reference's source text is stored.

usage:
    PYTHONPATH=/tmp/ntracer_oracle/build/lib.linux-x86_64-3.10 \
        python3 tools/gen_golden.py [--only NAME ...]

The polytope scenes come from the geometry half of the reference's
``scripts/polytope.py`` (executed in-memory with a stub ``pygame`` module, up
to the point where it would open a display).  Its output order is hash-order
dependent, so the *built* scene is captured; it is never regenerated.
"""
import argparse
import fractions
import math
import os
import struct
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("NTRACER_REF_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

from ntracer import NTracer, Material, ImageFormat, Channel, BlockingRenderer  # noqa: E402
import ntracer.render as R  # noqa: E402
from ntracer import wrapper as W  # noqa: E402

FRAMES = 160


# --------------------------------------------------------------------------
# cameras: the RotatingCamera of scripts/polytope.py:522-556, driven through
# the reference's own Camera/Matrix so the fixture holds the reference's fp32.
# --------------------------------------------------------------------------
def rotation_cameras(nt, cam_distance, frames=FRAMES, jitter0=True):
    n = nt.dimension
    jitter = nt.Vector((0, 0, 0) + (0.0001,) * (n - 3))
    cam = nt.Camera()
    v = nt.Vector.axis(2, cam_distance)
    if jitter0:
        v = v + jitter
    cam.translate(v)
    incr = 2 * math.pi / frames
    h = 1 / math.sqrt(n - 1)
    origins = np.zeros((frames, n), np.float32)
    axes = np.zeros((frames, n, n), np.float32)
    for f in range(frames):
        origins[f] = list(cam.origin)
        for i in range(n):
            axes[f, i] = list(cam.axes[i])
        a2 = cam.axes[0] * h + cam.axes[1] * h
        for i in range(n - 3):
            a2 += cam.axes[i + 3] * h
        cam.transform(nt.Matrix.rotation(cam.axes[2], a2, incr))
        cam.normalize()
        cam.origin = cam.axes[2] * cam_distance
    return origins, axes


def set_cam(nt, scene, origin, axes):
    cam = nt.Camera()
    cam.origin = nt.Vector(*[float(v) for v in origin])
    for i in range(nt.dimension):
        cam.axes[i] = nt.Vector(*[float(v) for v in axes[i]])
    scene.set_camera(cam)


def lattice(w, h, sx, sy, ox=0, oy=0):
    xs = np.arange(ox, w, sx, dtype=np.int32)
    ys = np.arange(oy, h, sy, dtype=np.int32)
    X, Y = np.meshgrid(xs, ys)
    return X.ravel(), Y.ravel()


def colors_at(scene, xs, ys, w, h):
    out = np.zeros((len(xs), 3), np.float32)
    for i, (x, y) in enumerate(zip(xs.tolist(), ys.tolist())):
        c = scene.calculate_color(x, y, w, h)
        out[i] = (c.r, c.g, c.b)
    return out


def render_bytes(scene, w, h, channels, pitch=0, reversed_=False, threads=0):
    fmt = ImageFormat(w, h, channels, pitch, reversed_)
    buf = bytearray(fmt.pitch * h)
    ok = BlockingRenderer(threads).render(buf, fmt, scene)
    assert ok
    return np.frombuffer(bytes(buf), np.uint8).reshape(h, fmt.pitch), fmt


RGBX8 = lambda: [Channel(8, 1, 0, 0), Channel(8, 0, 1, 0), Channel(8, 0, 0, 1), Channel(8, 0, 0, 0)]
RGB16 = lambda: [Channel(16, 1, 0, 0), Channel(16, 0, 1, 0), Channel(16, 0, 0, 1)]
RGBF32 = lambda: [Channel(32, 1, 0, 0, 0, True), Channel(32, 0, 1, 0, 0, True), Channel(32, 0, 0, 1, 0, True)]


def chan_table(channels):
    return np.array([[c.f_r, c.f_g, c.f_b, c.f_c, c.bit_size, 1 if c.tfloat else 0] for c in channels], np.float32)


# --------------------------------------------------------------------------
# BoxScene goldens (configs 1, 2, 3, 5)
# --------------------------------------------------------------------------
def gen_box():
    # config 1: BoxScene(3), 256x256, hypercube.py camera, RGBX8, 1 thread
    nt = NTracer(3)
    scene = nt.BoxScene()
    cam = nt.Camera()
    cam.translate(nt.Vector.axis(2, -5))
    scene.set_camera(cam)
    img, fmt = render_bytes(scene, 256, 256, RGBX8())
    xs, ys = lattice(256, 256, 5, 7)
    np.savez_compressed(
        os.path.join(OUT, "box_cfg1_n3_256.npz"),
        origin=np.array(list(cam.origin), np.float32),
        axes=np.array([list(cam.axes[i]) for i in range(3)], np.float32),
        fov=np.float32(scene.fov), image_rgbx8=img, xs=xs, ys=ys,
        colors=colors_at(scene, xs, ys, 256, 256))

    for n, (w, h), generic in ((3, (1920, 1080), False), (6, (1920, 1080), False),
                               (10, (4096, 4096), True), (6, (640, 480), True),
                               (5, (320, 200), False), (8, (320, 200), False), (12, (320, 200), True)):
        nt = NTracer(n, force_generic=generic)
        scene = nt.BoxScene()
        origins, axes = rotation_cameras(nt, -math.sqrt(n) * 4)
        frames = [0, 1, 17, 40, 93, 159]
        xs, ys = lattice(w, h, max(w // 96, 1) | 1, max(h // 54, 1) | 1, 3, 2)
        cols = np.zeros((len(frames), len(xs), 3), np.float32)
        for k, f in enumerate(frames):
            set_cam(nt, scene, origins[f], axes[f])
            cols[k] = colors_at(scene, xs, ys, w, h)
        # a dense patch straddling the silhouette for frame 17 (full rows)
        set_cam(nt, scene, origins[17], axes[17])
        rows = np.array([h // 2 - 1, h // 2, h // 3], np.int32)
        rx = np.arange(0, min(w, 1920), dtype=np.int32)
        dense = np.zeros((len(rows), len(rx), 3), np.float32)
        for k, y in enumerate(rows.tolist()):
            dense[k] = colors_at(scene, rx, np.full(len(rx), y, np.int32), w, h)
        tag = "box_n%d_%dx%d%s" % (n, w, h, "_generic" if generic and n <= 8 else "")
        np.savez_compressed(
            os.path.join(OUT, tag + ".npz"),
            origins=origins, axes=axes, fov=np.float32(scene.fov), frames=np.array(frames, np.int32),
            xs=xs, ys=ys, colors=cols, dense_rows=rows, dense_xs=rx, dense_colors=dense,
            width=np.int32(w), height=np.int32(h))
        print("wrote", tag)


# --------------------------------------------------------------------------
# pixel-packing goldens (render.cpp:396-466) on a small BoxScene(3) image
# --------------------------------------------------------------------------
def gen_packing():
    nt = NTracer(3)
    scene = nt.BoxScene()
    origins, axes = rotation_cameras(nt, -math.sqrt(3) * 4)
    set_cam(nt, scene, origins[23], axes[23])
    w, h = 67, 45   # not a multiple of anything
    formats = {
        "rgbx8": (RGBX8(), 0, False),
        "rgbx8_rev": (RGBX8(), 0, True),
        "rgb16": (RGB16(), 0, False),
        "rgb16_rev_pitch": (RGB16(), 67 * 6 + 10, True),
        "rgbf32": (RGBF32(), 0, False),
        "rgbf32_rev": (RGBF32(), 0, True),
        "rgb565": ([Channel(5, 1, 0, 0), Channel(6, 0, 1, 0), Channel(5, 0, 0, 1)], 0, False),
        "rgb888": ([Channel(8, 1, 0, 0), Channel(8, 0, 1, 0), Channel(8, 0, 0, 1)], 0, False),
        "bgr888_pitch": ([Channel(8, 0, 0, 1), Channel(8, 0, 1, 0), Channel(8, 1, 0, 0)], 67 * 3 + 5, False),
        "odd_1_7_13_31": ([Channel(1, 1, 0, 0), Channel(7, 0, 1, 0), Channel(13, 0, 0, 1), Channel(31, .3, .3, .3, .05)], 0, False),
        "gray10_pad": ([Channel(10, .299, .587, .114), Channel(3, 0, 0, 0, 1.0)], 0, False),
        "wide_16B": ([Channel(31, 1, 0, 0), Channel(31, 0, 1, 0), Channel(31, 0, 0, 1), Channel(32, .5, .5, 0, 0, True), Channel(3, 0, 0, 0, .5)], 0, False),
        "cross64": ([Channel(30, 1, 0, 0), Channel(30, 0, 1, 0), Channel(30, 0, 0, 1), Channel(30, 1, 1, 1, -0.5)], 0, True),
        "neg_mix": ([Channel(8, -1, 0, 0, 1), Channel(8, 2, -1, 0), Channel(8, 0, 0, 4, -1)], 0, False),
    }
    out = dict(origin=origins[23], axes=axes[23], fov=np.float32(scene.fov), width=np.int32(w), height=np.int32(h))
    xs, ys = lattice(w, h, 1, 1)
    out["colors"] = colors_at(scene, xs, ys, w, h).reshape(h, w, 3)
    names = []
    for name, (chans, pitch, rev) in formats.items():
        img, fmt = render_bytes(scene, w, h, chans, pitch, rev)
        out["fmt_%s_channels" % name] = chan_table(chans)
        out["fmt_%s_meta" % name] = np.array([fmt.pitch, 1 if rev else 0, fmt.bytes_per_pixel], np.int32)
        out["fmt_%s_image" % name] = img
        names.append(name)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "packing_box3.npz"), **out)
    print("wrote packing_box3")


# --------------------------------------------------------------------------
# composite scenes: flatten the reference's built tree
# --------------------------------------------------------------------------
KIND_BATCH, KIND_TRIANGLE, KIND_SOLID = 0, 1, 2


class Flattener:
    def __init__(self, nt):
        self.nt = nt
        self.n = nt.dimension
        self.nodes = []       # (axis | -1, split, left|item_start, right|item_count)
        self.items = []       # encoded (index << 2) | kind
        self.materials = []   # 10 floats each
        self.mat_ids = {}
        self.batches = {}     # id(obj) -> batch index
        self.batch_recs = []  # [B][rec]
        self.batch_mats = []
        self.tris = {}
        self.tri_recs = []
        self.tri_mats = []
        self.solids = {}
        self.solid_recs = []
        self.solid_types = []
        self.solid_mats = []
        self.keep = []        # keep python objects alive so id() stays unique

    def mat(self, m):
        key = (tuple(m.color), tuple(m.specular), m.opacity, m.reflectivity, m.specular_intensity, m.specular_exp)
        if key not in self.mat_ids:
            self.mat_ids[key] = len(self.materials)
            self.materials.append(list(m.color) + list(m.specular) +
                                  [m.opacity, m.reflectivity, m.specular_intensity, m.specular_exp])
        return self.mat_ids[key]

    def tri_rec(self, t):
        rec = [t.d] + list(t.face_normal) + list(t.p1)
        for e in t.edge_normals:
            rec += list(e)
        assert len(rec) == self.n * self.n + self.n + 1
        return rec

    def key(self, t):
        # primitives are exposed as fresh wrappers; identify them by content
        if isinstance(t, self.nt.base.Solid):
            return ("s", t.type, tuple(t.position), tuple(v for r in t.orientation for v in r))
        return ("t",) + tuple(self.tri_rec(t))   # full record: fans share p1/normal/d

    def add_item(self, p):
        base = self.nt.base
        if isinstance(p, base.TriangleBatch):
            tris = [p[i] for i in range(len(p))]
            k = tuple(self.key(t) for t in tris)
            if k not in self.batches:
                self.batches[k] = len(self.batch_recs)
                self.batch_recs.append([self.tri_rec(t) for t in tris])
                self.batch_mats.append([self.mat(t.material) for t in tris])
            return (self.batches[k] << 2) | KIND_BATCH
        if isinstance(p, base.Triangle):
            k = self.key(p)
            if k not in self.tris:
                self.tris[k] = len(self.tri_recs)
                self.tri_recs.append(self.tri_rec(p))
                self.tri_mats.append(self.mat(p.material))
            return (self.tris[k] << 2) | KIND_TRIANGLE
        assert isinstance(p, base.Solid)
        k = self.key(p)
        if k not in self.solids:
            self.solids[k] = len(self.solid_recs)
            rec = [v for r in p.orientation for v in r] + [v for r in p.inv_orientation for v in r] + list(p.position)
            self.solid_recs.append(rec)
            self.solid_types.append(int(p.type))
            self.solid_mats.append(self.mat(p.material))
        return (self.solids[k] << 2) | KIND_SOLID

    def add_node(self, node):
        """returns node index or -1"""
        if node is None:
            return -1
        base = self.nt.base
        idx = len(self.nodes)
        self.nodes.append(None)
        if isinstance(node, base.KDLeaf):
            start = len(self.items)
            for i in range(len(node)):
                self.items.append(self.add_item(node[i]))
            self.nodes[idx] = (-1, 0.0, start, len(node))
        else:
            l = self.add_node(node.left)
            r = self.add_node(node.right)
            self.nodes[idx] = (int(node.axis), float(node.split), l, r)
        return idx

    def arrays(self, scene):
        n = self.n
        root = self.add_node(scene.root)
        nodes = np.array(self.nodes, dtype=np.float64).reshape(-1, 4)
        d = dict(
            dimension=np.int32(n), batch_size=np.int32(self.nt.BATCH_SIZE), root=np.int32(root),
            node_axis=nodes[:, 0].astype(np.int32), node_split=nodes[:, 1].astype(np.float32),
            node_left=nodes[:, 2].astype(np.int32), node_right=nodes[:, 3].astype(np.int32),
            items=np.array(self.items, np.int32),
            batch_recs=np.array(self.batch_recs, np.float32).reshape(-1, self.nt.BATCH_SIZE, n * n + n + 1),
            batch_mats=np.array(self.batch_mats, np.int32).reshape(-1, self.nt.BATCH_SIZE),
            tri_recs=np.array(self.tri_recs, np.float32).reshape(-1, n * n + n + 1),
            tri_mats=np.array(self.tri_mats, np.int32),
            solid_recs=np.array(self.solid_recs, np.float32).reshape(-1, 2 * n * n + n),
            solid_types=np.array(self.solid_types, np.int32),
            solid_mats=np.array(self.solid_mats, np.int32),
            materials=np.array(self.materials, np.float32).reshape(-1, 10),
            aabb_start=np.array(list(scene.boundary.start), np.float32),
            aabb_end=np.array(list(scene.boundary.end), np.float32))
        return d


def scene_params(scene):
    pl = [(list(l.position), list(l.color)) for l in scene.point_lights]
    gl = [(list(l.direction), list(l.color)) for l in scene.global_lights]
    n = scene.dimension
    return dict(
        fov=np.float32(scene.fov), shadows=np.int32(scene.shadows), camera_light=np.int32(scene.camera_light),
        max_reflect_depth=np.int32(scene.max_reflect_depth), bg_gradient_axis=np.int32(scene.bg_gradient_axis),
        ambient=np.array(list(scene.ambient_color), np.float32),
        bg1=np.array(list(scene.bg1), np.float32), bg2=np.array(list(scene.bg2), np.float32),
        bg3=np.array(list(scene.bg3), np.float32),
        point_light_pos=np.array([p for p, _ in pl], np.float32).reshape(-1, n),
        point_light_color=np.array([c for _, c in pl], np.float32).reshape(-1, 3),
        global_light_dir=np.array([p for p, _ in gl], np.float32).reshape(-1, n),
        global_light_color=np.array([c for _, c in gl], np.float32).reshape(-1, 3))


def polytope_scene(schlafli):
    """Run the geometry half of the reference's scripts/polytope.py in memory."""
    src = open(os.path.join(REF_ROOT, "scripts", "polytope.py")).read()
    src = src.split("if args.output is not None:")[0]
    pg = types.ModuleType("pygame")
    pg.USEREVENT = 24
    pg.register_quit = lambda f: None
    sys.modules["pygame"] = pg
    sys.modules["ntracer.pygame_render"] = types.ModuleType("ntracer.pygame_render")
    sys.modules["ntracer.pygame_render"].PygameRenderer = object
    fractions.gcd = math.gcd
    argv = sys.argv
    sys.argv = ["polytope.py"] + schlafli
    g = {"__name__": "polytope_fixture"}
    hook = sys.excepthook
    try:
        exec(compile(src, "polytope.py", "exec"), g)
    finally:
        sys.argv = argv
        sys.excepthook = hook
    return g["nt"], g["scene"], g["cam_distance"]


def gen_polytope(name, schlafli, w, h, frames, step):
    nt, scene, cam_distance = polytope_scene(schlafli)
    fl = Flattener(nt)
    d = fl.arrays(scene)
    origins, axes = rotation_cameras(nt, cam_distance)
    xs, ys = lattice(w, h, step[0], step[1], 1, 2)
    cols = np.zeros((len(frames), len(xs), 3), np.float32)
    for k, f in enumerate(frames):
        set_cam(nt, scene, origins[f], axes[f])
        cols[k] = colors_at(scene, xs, ys, w, h)
    d.update(scene_params(scene))
    d.update(origins=origins, axes=axes, cam_distance=np.float32(cam_distance), frames=np.array(frames, np.int32),
             xs=xs, ys=ys, colors=cols, width=np.int32(w), height=np.int32(h))
    # one small full image (RGBX8) for byte-level parity of the whole frame loop
    set_cam(nt, scene, origins[frames[0]], axes[frames[0]])
    img, fmt = render_bytes(scene, 160, 90, RGBX8(), threads=3)
    d["image160x90_rgbx8"] = img
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, "nodes", len(d["node_axis"]), "items", len(d["items"]), "batches", len(d["batch_recs"]),
          "tris", len(d["tri_recs"]))


# --------------------------------------------------------------------------
# the hand-built scene of lib/ntracer/tests/test.py:303-363 (known answer)
# is restated in tests/ from the numbers in SURVEY.md; here we additionally
# capture what the reference answers for a fan of rays through that scene.
# --------------------------------------------------------------------------
def build_feature_scene():
    """3-D scene exercising lights, shadows, reflection, transparency, solids,
    unbatched triangles -- built with the reference's own builder."""
    import random
    rnd = random.Random(1234)
    nt = NTracer(3)
    mats = [Material((1, 0.5, 0.5)),
            Material((0.2, 0.9, 0.3), 1, 0.35, 0.8, 12, (1, 1, 0.8)),     # reflective
            Material((0.3, 0.4, 1.0), 0.45, 0, 1, 8),                     # transparent
            Material((0.9, 0.9, 0.9), 1, 0, 0, 8),                        # no specular
            Material((0.8, 0.6, 0.1), 0.7, 0.2, 0.5, 5, (0.5, 1, 1))]     # transparent + reflective
    protos = []
    # a floor of 2 big triangles
    V = nt.Vector
    def tri(a, b, c, m):
        protos.append(nt.TrianglePrototype([V(*a), V(*b), V(*c)], m))
    tri((-6, -2, -6), (6, -2, -6), (6, -2, 6), mats[1])
    tri((-6, -2, -6), (6, -2, 6), (-6, -2, 6), mats[1])
    # a cloud of small random triangles
    for i in range(41):
        c = [rnd.uniform(-3, 3), rnd.uniform(-1.5, 2.5), rnd.uniform(-3, 3)]
        pts = [[c[k] + rnd.uniform(-0.9, 0.9) for k in range(3)] for _ in range(3)]
        tri(pts[0], pts[1], pts[2], mats[i % len(mats)])
    # solids
    rot = nt.Matrix.rotation(V(1, 0, 0), V(0, 1, 0), 0.5) * nt.Matrix.rotation(V(0, 1, 0), V(0, 0, 1), 0.3) * nt.Matrix.scale(0.8)
    protos.append(nt.SolidPrototype(W.CUBE, V(1.5, -0.5, 0.5), rot, mats[0]))
    protos.append(nt.SolidPrototype(W.SPHERE, V(-1.6, 0.1, -0.4), nt.Matrix.scale(0.9), mats[4]))
    protos.append(nt.SolidPrototype(W.SPHERE, V(0.2, 1.4, 1.0), nt.Matrix.scale(0.5), mats[1]))
    scene = nt.build_composite_scene(protos)
    cam = nt.Camera()
    cam.translate(V(0.3, 0.8, -7))
    cam.transform(nt.Matrix.rotation(cam.axes[2], cam.axes[1], -0.12))
    cam.normalize()
    scene.set_camera(cam)
    return nt, scene, cam


def gen_feature_scene():
    nt, scene, cam = build_feature_scene()
    V = nt.Vector
    fl = Flattener(nt)
    d = fl.arrays(scene)
    origin = np.array(list(cam.origin), np.float32)
    axes = np.array([list(cam.axes[i]) for i in range(3)], np.float32)

    w, h = 96, 64
    xs, ys = lattice(w, h, 1, 1)
    variants = {}
    # v0: defaults (camera light only)
    variants["default"] = (scene_params(scene), colors_at(scene, xs, ys, w, h).reshape(h, w, 3))
    # v1: lights without shadows
    scene.add_light(nt.PointLight(V(3, 5, -4), (30, 28, 25)))
    scene.add_light(nt.PointLight(V(-4, 2, -2), (8, 12, 16)))
    scene.add_light(nt.GlobalLight(V(0.2, -1, 0.3).unit(), (0.5, 0.45, 0.4)))
    scene.set_ambient_color((0.05, 0.04, 0.06))
    scene.set_background((0.9, 0.8, 0.7), (0.1, 0.1, 0.2), (0, 0.3, 0.1), 1)
    variants["lights"] = (scene_params(scene), colors_at(scene, xs, ys, w, h).reshape(h, w, 3))
    # v2: + shadows
    scene.set_shadows(True)
    variants["shadows"] = (scene_params(scene), colors_at(scene, xs, ys, w, h).reshape(h, w, 3))
    # v3: shadows, no camera light, reflect depth 1, bg axis 0, fov 1.1
    scene.set_camera_light(False)
    scene.set_max_reflect_depth(1)
    scene.set_background((0.9, 0.8, 0.7), (0.1, 0.1, 0.2), (0, 0.3, 0.1), 0)
    scene.set_fov(1.1)
    variants["nocam_depth1"] = (scene_params(scene), colors_at(scene, xs, ys, w, h).reshape(h, w, 3))
    # v4: depth 0
    scene.set_max_reflect_depth(0)
    scene.set_camera_light(True)
    variants["depth0"] = (scene_params(scene), colors_at(scene, xs, ys, w, h).reshape(h, w, 3))

    d.update(origin=origin, axes=axes, width=np.int32(w), height=np.int32(h),
             variants=np.array(list(variants.keys())))
    for name, (params, cols) in variants.items():
        for k, v in params.items():
            d["%s__%s" % (name, k)] = v
        d["%s__colors" % name] = cols
    np.savez_compressed(os.path.join(OUT, "feature3d.npz"), **d)
    print("wrote feature3d nodes", len(d["node_axis"]), "items", len(d["items"]), "batches", len(d["batch_recs"]),
          "tris", len(d["tri_recs"]), "solids", len(d["solid_recs"]))


def gen_from_points():
    """Triangle.from_points / to_points of the reference on seeded random simplices (tracer.hpp:442-506)."""
    import random
    rnd = random.Random(99)
    out = {}
    dims = [3, 4, 5, 6, 8]
    for n in dims:
        nt = NTracer(n)
        mat = Material((1, 1, 1))
        pts = np.array([[[rnd.uniform(-3, 3) for _ in range(n)] for _ in range(n)] for _ in range(12)], np.float32)
        recs = []
        back = []
        for P in pts:
            t = nt.Triangle.from_points([nt.Vector(*[float(v) for v in p]) for p in P], mat)
            rec = [t.d] + list(t.face_normal) + list(t.p1)
            for e in t.edge_normals:
                rec += list(e)
            recs.append(rec)
            back.append([list(q) for q in t.to_points()])
        out["points_n%d" % n] = pts
        out["records_n%d" % n] = np.array(recs, np.float32)
        out["to_points_n%d" % n] = np.array(back, np.float32)
    out["dims"] = np.array(dims, np.int32)
    np.savez_compressed(os.path.join(OUT, "from_points.npz"), **out)
    print("wrote from_points")


def gen_pickles():
    """Pickles of the reference's picklable types (render.cpp:1094-1099,1197-1208,1696-1751; __reduce__ of
    Vector, Matrix, Triangle, TriangleBatch, Solid, AABB in ntracer_body.hpp): the pickle byte strings
    (protocol 2) and the values they were made from.  Data only."""
    import pickle
    import random
    rnd = random.Random(7)
    out = {}
    names = []

    def put(name, obj, values):
        out["pickle_" + name] = np.frombuffer(pickle.dumps(obj, 2), np.uint8)
        out["values_" + name] = np.asarray(values, np.float32)
        names.append(name)

    put("color", R.Color(0.25, 0.5, 0.75), [0.25, 0.5, 0.75])
    m = Material((1.0, 0.5, 0.25), 0.75, 0.125, 0.5, 12.0, (0.5, 0.25, 1.0))
    put("material", m, [1.0, 0.5, 0.25, 0.5, 0.25, 1.0, 0.75, 0.125, 0.5, 12.0])
    for n in (3, 5, 9):
        nt = NTracer(n)
        v = [rnd.uniform(-2, 2) for _ in range(n)]
        put("vector%d" % n, nt.Vector(*v), v)
        mv = [[rnd.uniform(-2, 2) for _ in range(n)] for _ in range(n)]
        put("matrix%d" % n, nt.Matrix(mv), mv)
        put("aabb%d" % n, nt.AABB(nt.Vector(*[-1.0 - i for i in range(n)]), nt.Vector(*[2.0 + i for i in range(n)])),
            [[-1.0 - i for i in range(n)], [2.0 + i for i in range(n)]])
        pts = [[rnd.uniform(-3, 3) for _ in range(n)] for _ in range(n)]
        t = nt.Triangle.from_points([nt.Vector(*q) for q in pts], m)
        rec = [list(t.p1), list(t.face_normal)] + [list(e) for e in t.edge_normals]
        put("triangle%d" % n, t, rec)
        tris = []
        for k in range(4):
            q = [[rnd.uniform(-3, 3) for _ in range(n)] for _ in range(n)]
            tris.append(nt.Triangle.from_points([nt.Vector(*x) for x in q], m))
        tb = nt.TriangleBatch(tris)
        put("batch%d" % n, tb, [[list(x.p1), list(x.face_normal)] + [list(e) for e in x.edge_normals] for x in tris])
        rot = nt.Matrix.rotation(nt.Vector.axis(0), nt.Vector.axis(1), 0.3) * nt.Matrix.scale(1.5)
        so = nt.Solid(W.CUBE if n != 5 else W.SPHERE, nt.Vector(*[0.5 * i for i in range(n)]), rot, m)
        put("solid%d" % n, so, list(so.orientation.values) + list(so.position))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "pickles.npz"), **out)
    print("wrote pickles", names)


def gen_lit12():
    """A 12-D scene through the reference's generic module (`tracern`, var_geometry.hpp -- there is no tracer12): the 13
    facets of a regular 12-simplex (reflective and plain materials alternating) with a Solid cube and a Solid sphere next
    to it, a point light, a global light, shadows on, reflection depth 2.  Built by the reference's own builder."""
    n = 12
    nt = NTracer(n)
    V = nt.Vector
    mats = [Material((1, 0.5, 0.5)), Material((0.3, 0.8, 0.4), 1, 0.3, 0.7, 10, (1, 1, 0.7)), Material((0.85, 0.85, 0.9), 1, 0, 0, 8)]
    pts = [[1.0 if k == i else 0.0 for k in range(n)] for i in range(n)]
    pts.append([(1 - math.sqrt(n + 1)) / n] * n)
    centre = [sum(p[k] for p in pts) / (n + 1) for k in range(n)]
    pts = [[2.5 * (p[k] - centre[k]) for k in range(n)] for p in pts]
    protos = []
    for skip in range(n + 1):
        protos.append(nt.TrianglePrototype([V(*p) for i, p in enumerate(pts) if i != skip], mats[skip % 2]))
    ax = lambda i: V.axis(i, 1)
    rot = nt.Matrix.rotation(ax(0), ax(1), 0.4) * nt.Matrix.rotation(ax(2), ax(7), 0.3) * nt.Matrix.scale(0.9)
    protos.append(nt.SolidPrototype(W.CUBE, V(*([2.6, -1.2, 0.4] + [0.1] * (n - 3))), rot, mats[2]))
    protos.append(nt.SolidPrototype(W.SPHERE, V(*([-2.4, 1.5, -0.3] + [0.0] * (n - 3))), nt.Matrix.scale(1.1), mats[1]))
    scene = nt.build_composite_scene(protos)
    scene.add_light(nt.PointLight(V(*([5.0, 6.0, -7.0, 2.0, 1.0, -1.0, 0.5, 0.0, 2.0, -3.0, 1.0, 0.5])), (9e10, 8e10, 7e10)))
    scene.add_light(nt.GlobalLight(V(*([0.2, -0.9, 0.3, 0.1, 0.0, 0.1, -0.1, 0.0, 0.05, 0.0, 0.1, -0.05])).unit(), (0.4, 0.4, 0.5)))
    scene.set_ambient_color((0.02, 0.02, 0.03))
    scene.set_shadows(True)
    scene.set_max_reflect_depth(2)
    fl = Flattener(nt)
    d = fl.arrays(scene)
    cam_distance = -2.5 * 4
    origins, axes = rotation_cameras(nt, cam_distance, frames=40)
    w, h = 160, 100
    frames = [0, 7, 19, 33]
    xs, ys = lattice(w, h, 3, 2, 1, 1)
    cols = np.zeros((len(frames), len(xs), 3), np.float32)
    for k, f in enumerate(frames):
        set_cam(nt, scene, origins[f], axes[f])
        cols[k] = colors_at(scene, xs, ys, w, h)
    d.update(scene_params(scene))
    d.update(origins=origins, axes=axes, cam_distance=np.float32(cam_distance), frames=np.array(frames, np.int32),
             xs=xs, ys=ys, colors=cols, width=np.int32(w), height=np.int32(h))
    np.savez_compressed(os.path.join(OUT, "lit12_n12.npz"), **d)
    print("wrote lit12_n12 nodes", len(d["node_axis"]), "items", len(d["items"]), "batches", len(d["batch_recs"]), "tris", len(d["tri_recs"]),
          "solids", len(d["solid_recs"]), "hit fraction", float((np.abs(cols[..., 0] - cols[..., 1]) > 1e-6).mean()))


def gen_feature_n(n, name, w=160, h=100, frames=(0, 9, 21, 34), depth=3):
    """A scene with everything the composite path has, through the reference's tracer<n> module (n = 5) or its generic
    run-time-n one (n = 11): facets of an n-simplex in opaque, reflective, transparent and transparent + reflective
    materials, a cloud of small simplices, a Solid cube and a Solid sphere (one transparent), a point light, a global light,
    shadows, reflection.  Built by the reference's own builder."""
    import random
    nt = NTracer(n)
    V = nt.Vector
    mats = [Material((1, 0.5, 0.5)), Material((0.3, 0.8, 0.4), 1, 0.3, 0.7, 10, (1, 1, 0.7)), Material((0.3, 0.4, 1.0), 0.5, 0, 1, 8),
            Material((0.8, 0.6, 0.1), 0.7, 0.2, 0.5, 5, (0.5, 1, 1)), Material((0.85, 0.85, 0.9), 1, 0, 0, 8)]
    pts = [[1.0 if k == i else 0.0 for k in range(n)] for i in range(n)]
    pts.append([(1 - math.sqrt(n + 1)) / n] * n)
    centre = [sum(p[k] for p in pts) / (n + 1) for k in range(n)]
    # (stretched a little differently along every axis: a regular simplex is symmetric under swapping the axes the camera
    # never turns into, and two of its facets would then tie along whole regions of rays)
    pts = [[2.2 * (1 + 0.04 * k) * (p[k] - centre[k]) for k in range(n)] for p in pts]
    protos = []
    for skip in range(n + 1):
        protos.append(nt.TrianglePrototype([V(*p) for i, p in enumerate(pts) if i != skip], mats[skip % 4]))
    rnd = random.Random(77)
    for i in range(9):                        # a cloud of small simplices around it
        c = [rnd.uniform(-2.5, 2.5) for _ in range(n)]
        protos.append(nt.TrianglePrototype([V(*[c[k] + rnd.uniform(-0.8, 0.8) for k in range(n)]) for _ in range(n)], mats[i % 5]))
    ax = lambda i: V.axis(i, 1)
    pad = [0.0] * (n - 5)
    rot = nt.Matrix.rotation(ax(0), ax(1), 0.4) * nt.Matrix.rotation(ax(2), ax(4), 0.3) * nt.Matrix.scale(0.8)
    protos.append(nt.SolidPrototype(W.CUBE, V(2.3, -1.0, 0.4, 0.1, -0.2, *pad), rot, mats[4]))
    protos.append(nt.SolidPrototype(W.SPHERE, V(-2.1, 1.3, -0.3, 0.2, 0.0, *pad), nt.Matrix.scale(0.9), mats[3]))
    scene = nt.build_composite_scene(protos)
    # (a point light's strength falls with distance^(n-1): its colour is scaled so that it matters at this distance)
    ppos = [5.0, 6.0, -7.0, 2.0, 1.0] + pad
    pd = math.sqrt(sum(v * v for v in ppos))
    k = 2.27 * pd ** (n - 1)
    scene.add_light(nt.PointLight(V(*ppos), (k, 0.93 * k, 0.83 * k)))
    scene.add_light(nt.GlobalLight(V(0.2, -0.9, 0.3, 0.1, 0.05, *pad).unit(), (0.4, 0.4, 0.5)))
    scene.set_ambient_color((0.03, 0.03, 0.04))
    scene.set_shadows(True)
    scene.set_max_reflect_depth(depth)
    fl = Flattener(nt)
    d = fl.arrays(scene)
    cam_distance = -2.2 * 4
    origins, axes = rotation_cameras(nt, cam_distance, frames=40)
    frames = list(frames)
    xs, ys = lattice(w, h, 2, 2, 1, 0)
    cols = np.zeros((len(frames), len(xs), 3), np.float32)
    for k, f in enumerate(frames):
        set_cam(nt, scene, origins[f], axes[f])
        cols[k] = colors_at(scene, xs, ys, w, h)
    d.update(scene_params(scene))
    d.update(origins=origins, axes=axes, cam_distance=np.float32(cam_distance), frames=np.array(frames, np.int32),
             xs=xs, ys=ys, colors=cols, width=np.int32(w), height=np.int32(h))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, "nodes", len(d["node_axis"]), "items", len(d["items"]), "batches", len(d["batch_recs"]), "tris", len(d["tri_recs"]),
          "solids", len(d["solid_recs"]), "hit fraction", float((np.abs(cols[..., 0] - cols[..., 1]) > 1e-6).mean()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    jobs = {
        "box": gen_box,
        "packing": gen_packing,
        "cell600": lambda: gen_polytope("cell600_n4", ["3", "3", "5"], 640, 360, [0, 5, 33, 77, 121], (9, 7)),
        "cell120": lambda: gen_polytope("cell120_n4", ["5/2", "3", "3"], 1920, 1080, [0, 11, 52, 97, 140], (37, 29)),
        "feature": gen_feature_scene,
        "from_points": gen_from_points,
        "pickles": gen_pickles,
        # a 10-D simplex {3,3,3,3,3,3,3,3,3}: composite scene through the generic (var_geometry) module
        "simplex10": lambda: gen_polytope("simplex10_n10", ["3"] * 9, 320, 200, [0, 9, 47, 120], (5, 3)),
        # a 5-D cross-polytope-like {3,3,3,4}: 32 facets, fixed<5> module
        "orthoplex5": lambda: gen_polytope("orthoplex5_n5", ["3", "3", "3", "4"], 320, 200, [0, 9, 47, 120], (5, 3)),
        # 7-D and 9-D simplices: the reference's tracer7 module and its generic one (n = 9 has no specialised module there)
        "simplex7": lambda: gen_polytope("simplex7_n7", ["3"] * 6, 320, 200, [0, 9, 47, 120], (5, 3)),
        "simplex9": lambda: gen_polytope("simplex9_n9", ["3"] * 8, 320, 200, [0, 9, 47, 120], (5, 3)),
        "lit12": gen_lit12,
        # (feature5_n5 was captured with the point light (3e4, 2.8e4, 2.5e4); regenerating it changes that light a little)
        "feature5": lambda: gen_feature_n(5, "feature5_n5"),
        "feature11": lambda: gen_feature_n(11, "feature11_n11", frames=(0, 9, 21)),
        "feature16": lambda: gen_feature_n(16, "feature16_n16", frames=(3, 17)),
    }
    for k, f in jobs.items():
        if a.only is None or k in a.only:
            f()

"""The oracle (oracle/ntracer_oracle.c) against every golden vector captured from the compiled reference
and the reference's own known-answer test.  CPU only.  This is what pins the oracle."""
import numpy as np
import pytest

import fixtures as fx
import oracle_binding as ob

TOL = 1e-4      # north_star: max per-channel |delta pixel| < 1e-4


@pytest.mark.parametrize("name", fx.BOX_FIXTURES)
def test_box_colors_match_reference(name):
    g = fx.load(name)
    n = g["origins"].shape[1]
    w, h = int(g["width"]), int(g["height"])
    bad = 0
    total = 0
    worst_smooth = 0.0
    for k, f in enumerate(g["frames"]):
        sc = ob.OracleScene(n, g["origins"][f], g["axes"][f], float(g["fov"]))
        d = np.abs(sc.colors_at(g["xs"], g["ys"], w, h) - g["colors"][k]).max(axis=1)
        bad += int((d > TOL).sum())
        total += len(d)
        worst_smooth = max(worst_smooth, float(d[d <= TOL].max()))
    sc = ob.OracleScene(n, g["origins"][17], g["axes"][17], float(g["fov"]))
    for k, y in enumerate(g["dense_rows"]):
        xs = g["dense_xs"]
        d = np.abs(sc.colors_at(xs, np.full(len(xs), y), w, h) - g["dense_colors"][k]).max(axis=1)
        bad += int((d > TOL).sum())
        total += len(d)
    # the reference is built -ffast-math: a ray within ~1e-6 of a cube edge may pick the other face
    assert bad <= max(1, total // 20000), "%d of %d samples differ from the reference" % (bad, total)
    assert worst_smooth < 1e-6


def test_box_config1_bytes():
    """config 1: BoxScene(3), 256x256, RGBX8, single thread -- whole image, byte for byte."""
    g = fx.load("box_cfg1_n3_256")
    sc = ob.OracleScene(3, g["origin"], g["axes"], float(g["fov"]))
    img = sc.render(256, 256, fx.RGBX8, threads=0)
    assert np.array_equal(img, g["image_rgbx8"])
    assert np.abs(sc.colors_at(g["xs"], g["ys"], 256, 256) - g["colors"]).max() < 1e-6


def test_pixel_packing_all_formats_bit_exact():
    """process_pixel (render.cpp:396-466): 14 channel layouts incl. 31-bit channels, a 64-bit-boundary
    crossing, float channels, reversed byte order and padded pitch."""
    g = fx.load("packing_box3")
    w, h = int(g["width"]), int(g["height"])
    sc = ob.OracleScene(3, g["origin"], g["axes"], float(g["fov"]))
    assert len(g["names"]) >= 14
    for name in g["names"]:
        pitch, rev, bpp = [int(v) for v in g["fmt_%s_meta" % name]]
        img = sc.render(w, h, ob.channels_from_table(g["fmt_%s_channels" % name]), pitch, bool(rev))
        assert np.array_equal(img[:, :w * bpp], g["fmt_%s_image" % name][:, :w * bpp]), name


def test_pack_pixel_edge_values():
    # clamp, rounding (lround: half away from zero) and MSB-first packing
    assert ob.pack_pixel([2.0, -1.0, 0.5], fx.RGBX8) == bytes([255, 0, 128, 0])
    assert ob.pack_pixel([1.0, 0.0, 0.0], [(5, 1, 0, 0), (6, 0, 1, 0), (5, 0, 0, 1)]) == bytes([0xF8, 0x00])
    assert ob.pack_pixel([0.0, 1.0, 1.0], [(5, 1, 0, 0), (6, 0, 1, 0), (5, 0, 0, 1)], True) == bytes([0xFF, 0x07])
    assert ob.pack_pixel([0.25, 0, 0], [(32, 1, 0, 0, 0, True)]) == bytes([0x3E, 0x80, 0, 0])   # big-endian float
    assert ob.pack_pixel([float("nan"), 0, 0], [(8, 1, 0, 0)]) == bytes([0])                     # SSE max(NaN,0) = 0


@pytest.mark.parametrize("name", ["cell600_n4", "cell120_n4", "orthoplex5_n5", "simplex7_n7", "simplex9_n9", "simplex10_n10"])
def test_polytope_scene_matches_reference(name):
    g = fx.load(name)
    n = int(g["dimension"])
    w, h = int(g["width"]), int(g["height"])
    flat = fx.flat_of(g)
    params = fx.params_of(g)
    for k, f in enumerate(g["frames"]):
        sc = ob.OracleScene(n, g["origins"][f], g["axes"][f], flat=flat, params=params)
        d = np.abs(sc.colors_at(g["xs"], g["ys"], w, h) - g["colors"][k])
        assert d.max() < 1e-5, (name, int(f), float(d.max()))
    f0 = g["frames"][0]
    sc = ob.OracleScene(n, g["origins"][f0], g["axes"][f0], flat=flat, params=params)
    img = sc.render(160, 90, fx.RGBX8, threads=3)
    assert np.array_equal(img, g["image160x90_rgbx8"])


def test_cell120_work_counters():
    """The byte model of DESIGN.md / SURVEY 8d is fed by these counters; pin their order of magnitude
    to the survey's probe (32.3 branches, 4.65 leaves, 49.8 batches per primary ray on frame 0)."""
    g = fx.load("cell120_n4")
    sc = ob.OracleScene(4, g["origins"][0], g["axes"][0], flat=fx.flat_of(g), params=fx.params_of(g))
    _, c = sc.colors_at(g["xs"], g["ys"], 1920, 1080, counters=True)
    rays = c["rays"]
    assert rays == len(g["xs"])
    assert 25 < c["branches"] / rays < 40
    assert 3.5 < c["leaves"] / rays < 6
    assert 35 < c["batch_tests"] / rays < 65
    assert 0.25 < c["hits"] / rays < 0.45


def test_feature_scene_matches_reference():
    """Lights, shadows (incl. the _occludes far-child quirk), reflection, transparency, solids and
    unbatched triangles.  The reference aliases o_hit.normal as scratch (see oracle header): the
    oracle reproduces that; the few remaining outliers are rays that start ON a surface because of
    that aliasing (self-intersection decided by the last bit)."""
    g = fx.load("feature3d")
    w, h = int(g["width"]), int(g["height"])
    ys, xs = np.mgrid[0:h, 0:w]
    flat = fx.flat_of(g)
    for v in g["variants"]:
        sc = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=fx.params_of(g, "%s__" % v))
        c = sc.colors_at(xs.ravel(), ys.ravel(), w, h).reshape(h, w, 3)
        d = np.abs(c - g["%s__colors" % v]).max(axis=2)
        assert (d > TOL).sum() <= 0.005 * d.size, (str(v), int((d > TOL).sum()))
        assert np.median(d) < 1e-6


def test_twelve_dimensional_lit_scene_matches_the_generic_reference_module():
    """lit12_n12: the reference's generic `tracern` (var_geometry.hpp; there is no tracer12) on the facets of a 12-simplex,
    a Solid cube and sphere, a point and a global light, shadows, reflection depth 2."""
    g = fx.load("lit12_n12")
    flat = fx.flat_of(g)
    assert len(flat["solid_recs"]) == 2 and len(flat["tri_recs"]) == 1 and len(flat["batch_recs"]) == 3
    p = fx.params_of(g)
    bad = total = 0
    for k, f in enumerate(g["frames"]):
        c = ob.OracleScene(12, g["origins"][f], g["axes"][f], flat=flat, params=p).colors_at(g["xs"], g["ys"], 160, 100)
        d = np.abs(c - g["colors"][k]).max(axis=1)
        bad += int((d > TOL).sum())
        total += len(d)
        assert np.median(d) < 1e-6
    assert bad <= 0.005 * total, bad


def test_five_dimensional_feature_scene_matches_the_reference():
    """feature5_n5, rendered by the reference's tracer5: opaque / reflective / transparent / transparent + reflective
    simplices (batched and loose), a Solid cube, a transparent Solid sphere, point + global light, shadows, reflection depth 3.
    The default mode (o_hit.normal aliasing reproduced) agrees on every sample; the clean mode does not."""
    g = fx.load("feature5_n5")
    flat = fx.flat_of(g)
    assert len(flat["solid_recs"]) == 2 and len(flat["tri_recs"]) == 3 and (np.asarray(flat["materials"])[:, 6] < 1).sum() == 2
    p = fx.params_of(g)
    differ_clean = 0
    for k, f in enumerate(g["frames"]):
        c = ob.OracleScene(5, g["origins"][f], g["axes"][f], flat=flat, params=p).colors_at(g["xs"], g["ys"], 160, 100)
        assert np.abs(c - g["colors"][k]).max() < TOL, int(f)
        cc = ob.OracleScene(5, g["origins"][f], g["axes"][f], flat=flat, params=p, clean_normals=True).colors_at(g["xs"], g["ys"], 160, 100)
        differ_clean += int((np.abs(cc - g["colors"][k]).max(axis=1) > TOL).sum())
    assert differ_clean > 10


def test_eleven_dimensional_feature_scene_matches_the_reference():
    """feature11_n11: the same kind of scene through the reference's generic run-time-n module (var_geometry) -- the
    default mode agrees with it on every sample in eleven dimensions too."""
    g = fx.load("feature11_n11")
    flat = fx.flat_of(g)
    p = fx.params_of(g)
    differ_clean = 0
    for k, f in enumerate(g["frames"]):
        c = ob.OracleScene(11, g["origins"][f], g["axes"][f], flat=flat, params=p).colors_at(g["xs"], g["ys"], 160, 100)
        assert np.abs(c - g["colors"][k]).max() < TOL, int(f)
        cc = ob.OracleScene(11, g["origins"][f], g["axes"][f], flat=flat, params=p, clean_normals=True).colors_at(g["xs"], g["ys"], 160, 100)
        differ_clean += int((np.abs(cc - g["colors"][k]).max(axis=1) > TOL).sum())
    assert differ_clean > 0


def test_sixteen_dimensional_feature_scene_matches_the_reference():
    """feature16_n16: the same kind of scene in sixteen dimensions (the reference's generic module again)."""
    g = fx.load("feature16_n16")
    flat = fx.flat_of(g)
    p = fx.params_of(g)
    for k, f in enumerate(g["frames"]):
        c = ob.OracleScene(16, g["origins"][f], g["axes"][f], flat=flat, params=p).colors_at(g["xs"], g["ys"], 160, 100)
        assert np.abs(c - g["colors"][k]).max() < TOL, int(f)


def test_clean_mode_only_differs_where_the_alias_bites():
    g = fx.load("cell600_n4")
    f = g["frames"][1]
    a = ob.OracleScene(4, g["origins"][f], g["axes"][f], flat=fx.flat_of(g)).colors_at(g["xs"], g["ys"], 640, 360)
    b = ob.OracleScene(4, g["origins"][f], g["axes"][f], flat=fx.flat_of(g), clean_normals=True).colors_at(g["xs"], g["ys"], 640, 360)
    assert np.array_equal(a, b)          # no solids, no transparency: identical
    g = fx.load("feature3d")
    ys, xs = np.mgrid[0:64, 0:96]
    flat = fx.flat_of(g, opaque=True)
    p = fx.params_of(g, "lights__")
    a = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p).colors_at(xs.ravel(), ys.ravel(), 96, 64)
    b = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p, clean_normals=True).colors_at(xs.ravel(), ys.ravel(), 96, 64)
    differ = (np.abs(a - b).max(axis=1) > TOL).sum()
    assert 0 < differ < 0.05 * len(a)    # a missed Solid-cube test scribbles on the hit's normal.origin


def test_reference_known_answer_kdtree():
    """lib/ntracer/tests/test.py:303-363: one ray, exactly one hit, on primitives[4], batch_index -1."""
    ka = fx.known_answer()
    flat = fx.known_answer_flat(ka)
    sc = ob.OracleScene(3, [0, 0, 0], np.eye(3), float(ka["fov"]), flat=flat)
    r = sc.kd_intersects(ka["ray"]["origin"], ka["ray"]["direction"])
    assert r is not None
    assert r["n_transparent"] + 1 == ka["expect"]["n_hits"]
    assert (r["kind"], r["index"], r["lane"]) == (1, ka["expect"]["primitive_index"], ka["expect"]["batch_index"])


def test_occludes_far_child_quirk():
    """KDBranch(0, 0, leaf[A], leaf[B]).occludes((-2,0,-.2),(1,0,0),10) is False in the
    reference although B blocks the ray at t=3 (tracer.hpp:1298), while intersects() finds B at 3.0."""
    n = 3

    def tri(p1, fn, e):
        p1 = np.asarray(p1, np.float32)
        fn = np.asarray(fn, np.float32)
        return [-float(np.dot(fn, p1))] + list(fn) + list(p1) + list(np.asarray(e, np.float32).ravel())

    # B: triangle in the plane x=+1 around (1,0,0); A: in x=-1 but far away in y
    # edge normals for a right triangle with legs 4: p1=(x,-1,-1): points p1, p1+(0,4,0), p1+(0,0,4)
    B = tri([1, -1, -1], [1, 0, 0], [[0, -0.25, 0], [0, 0, -0.25]])
    A = tri([-1, 5, -1], [1, 0, 0], [[0, -0.25, 0], [0, 0, -0.25]])
    flat = dict(root=0, node_axis=np.array([0, -1, -1], np.int32), node_split=np.array([0, 0, 0], np.float32),
                node_left=np.array([1, 0, 1], np.int32), node_right=np.array([2, 1, 1], np.int32),
                items=np.array([(0 << 2) | 1, (1 << 2) | 1], np.int32),
                batch_recs=np.zeros((0, 4, 13), np.float32), batch_mats=np.zeros((0, 4), np.int32),
                tri_recs=np.asarray([A, B], np.float32), tri_mats=np.zeros(2, np.int32),
                solid_recs=np.zeros((0, 21), np.float32), solid_types=np.zeros(0, np.int32),
                solid_mats=np.zeros(0, np.int32), materials=np.asarray([[1, 1, 1, 1, 1, 1, 1, 0, 1, 8]], np.float32),
                aabb_start=np.array([-3, -3, -3], np.float32), aabb_end=np.array([3, 9, 3], np.float32), batch_size=4)
    sc = ob.OracleScene(3, [0, 0, 0], np.eye(3), flat=flat)
    hit = sc.kd_intersects([-2, 0, -.2], [1, 0, 0])
    assert hit is not None and hit["index"] == 1 and abs(hit["dist"] - 3.0) < 1e-6
    for dist in (10.0, 1.5, 3.4e38):
        assert sc.kd_occludes([-2, 0, -.2], [1, 0, 0], dist) is False
    # from the other side the blocker is in the near cell and IS seen
    assert sc.kd_occludes([2, 0, -.2], [-1, 0, 0], 10.0) is True


def test_threaded_render_is_deterministic():
    g = fx.load("cell600_n4")
    f = g["frames"][2]
    sc = ob.OracleScene(4, g["origins"][f], g["axes"][f], flat=fx.flat_of(g))
    a = sc.render(97, 65, fx.RGB16, threads=0)
    b = sc.render(97, 65, fx.RGB16, threads=5)
    assert np.array_equal(a, b)


def test_persistent_renderer_reuses_its_workers_across_frames():
    """blocking_renderer (render.cpp:769-851): the workers sleep on start_cond between frames.  Many frames through one
    renderer -- one at a time and inside one C call -- give the bytes of fresh single-frame renders."""
    g = fx.load("box_n6_1920x1080")
    sc = ob.OracleScene(6, g["origins"][0], g["axes"][0])
    r = ob.OracleRenderer(3)
    assert r.threads == 4
    w, h = 131, 77
    for f in (0, 11, 97, 11):
        sc.set_camera(g["origins"][f], g["axes"][f])
        assert np.array_equal(r.render(sc, w, h, fx.RGBX8), sc.render(w, h, fx.RGBX8, threads=0)), f
    buf, secs = r.render_frames(sc, w, h, fx.RGBX8, g["origins"][:40], g["axes"][:40], 100)
    assert len(secs) == 100 and (secs > 0).all()
    sc.set_camera(g["origins"][99 % 40], g["axes"][99 % 40])
    assert np.array_equal(buf, sc.render(w, h, fx.RGBX8, threads=0))
    # the time limit stops the loop early
    _, secs = r.render_frames(sc, w, h, fx.RGBX8, g["origins"][:40], g["axes"][:40], 100000, max_seconds=0.05)
    assert 1 <= len(secs) < 100000
    r.close()
    # a renderer with no workers draws everything on the caller
    r0 = ob.OracleRenderer(0)
    assert r0.threads == 1
    assert np.array_equal(r0.render(sc, w, h, fx.RGBX8), sc.render(w, h, fx.RGBX8, threads=2))
    r0.close()


@pytest.mark.parametrize("name", ["cell600_n4", "orthoplex5_n5", "simplex10_n10"])
def test_pruned_walk_is_pixel_identical_on_the_oracle(name):
    """nto_scene.prune_beyond_hit (what the HIP kernels do unless strict_reference is set) against the
    reference's walk: same packed frame, fewer batch tests."""
    g = fx.load(name)
    n = int(g["dimension"])
    flat = fx.flat_of(g)
    f = int(g["frames"][1])
    res = []
    for prune in (False, True):
        buf, cnt = ob.OracleScene(n, g["origins"][f], g["axes"][f], flat=flat, prune=prune).render(160, 90, fx.RGBF32, threads=4, counters=True)
        res.append((buf, cnt))
    assert np.array_equal(res[0][0], res[1][0])
    assert res[1][1]["batch_tests"] <= res[0][1]["batch_tests"]
    assert res[1][1]["leaves"] <= res[0][1]["leaves"]


def test_pruned_walk_identical_on_the_feature_scene():
    g = fx.load("feature3d")
    flat = fx.flat_of(g)
    w, h = int(g["width"]), int(g["height"])
    flat = fx.flat_of(g, opaque=True)                  # transparency always walks strictly
    for v in g["variants"]:
        p = fx.params_of(g, "%s__" % v)
        a = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p, clean_normals=True).render(w, h, fx.RGBF32, threads=4)
        b = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p, clean_normals=True, prune=True).render(w, h, fx.RGBF32, threads=4)
        assert np.array_equal(a, b), v

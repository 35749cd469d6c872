"""Regular polytopes from Schläfli symbols on our side.  The generator is a different
construction from the reference's scripts/polytope.py (Wythoff orbits instead of facet propagation), so it is
checked against what the reference built: the fixtures' scenes were flattened from the reference's own
``hull()`` + ``build_composite_scene``; the generated polytope, partitioned by OUR builder, must give the oracle
the same colours on the same cameras (the reference inflates facets by 1e-5 per propagation step, so a handful
of silhouette samples may differ)."""
import math

import numpy as np
import pytest

import fixtures as fx
import ntracer_amd
import oracle_binding as ob
from ntracer_amd import polytope, tracern
from ntracer_amd.wrapper import NTracer

MAT = ntracer_amd.Material((1, .5, .5))


@pytest.mark.parametrize("symbol,verts,facets,simplices", [
    (["5"], 5, 5, 3), (["5/2"], 5, 5, 5), (["4", "3"], 8, 6, 12), (["3", "5"], 12, 20, 20),
    (["5/2", "5"], 12, 12, 60), (["5", "5/2"], 12, 12, 36), (["3", "5/2"], 12, 20, 20), (["5/2", "3"], 20, 12, 60),
    (["3", "4", "3"], 24, 24, 96), (["3", "3", "5"], 120, 600, 600), (["5", "3", "3"], 600, 120, 3240),
    (["5/2", "3", "3"], 600, 120, 7200), (["3", "3", "3", "4"], 10, 32, 32), (["3"] * 9, 11, 11, 11),
    (["4", "3", "3", "3"], 32, 10, 240)])
def test_element_counts(symbol, verts, facets, simplices):
    p = polytope.RegularPolytope(symbol)
    assert len(p.vertices) == verts
    assert len(p.faces[p.rank - 1]) == facets
    s = p.simplices(0)                                 # no subdivision: the raw tessellation
    n = p.dimension
    assert s.shape == (simplices, n, n)
    # every vertex on the circumsphere, every 2-face edge at distance 1 from its face centre
    r = np.linalg.norm(p.vertices, axis=1)
    assert np.allclose(r, p.circumradius(), rtol=1e-9)
    if p.rank >= 3:
        f2 = p.faces[2][0]
        c2 = p.vertices[list(f2)].mean(axis=0)
        for e in p._subfaces(2, f2):
            assert abs(np.linalg.norm(p.vertices[list(e)].mean(axis=0) - c2) - 1.0) < 1e-9
    # simplices are non-degenerate
    vol = np.abs(np.linalg.det((s[:, 1:, :] - s[:, :1, :]).astype(np.float64) @ np.swapaxes(s[:, 1:, :] - s[:, :1, :], 1, 2).astype(np.float64)))
    assert (vol > 1e-9).all()


def test_symbol_parsing_and_rejections():
    assert polytope.schlafli_component("5/2") == polytope.fractions.Fraction(5, 2)
    for bad in ("2", "5/0", "5/5", "6/2"):
        with pytest.raises(ValueError):
            polytope.schlafli_component(bad)
    with pytest.raises(ValueError, match="can't be folded inward"):
        polytope.RegularPolytope(["6", "3"])           # a plane tiling
    with pytest.raises(ValueError):
        polytope.RegularPolytope(["4", "4"])
    assert polytope.is_hypercube([polytope.schlafli_component(c) for c in ("4", "3", "3")])
    nt, scene, dist = polytope.build_scene(["4", "3", "3", "3", "3"])
    assert isinstance(scene, tracern.BoxScene) and scene.dimension == 6 and abs(dist + math.sqrt(6) * 4) < 1e-12


@pytest.mark.parametrize("name,symbol", [("cell600_n4", ["3", "3", "5"]), ("orthoplex5_n5", ["3", "3", "3", "4"]),
                                         ("simplex10_n10", ["3"] * 9), ("cell120_n4", ["5/2", "3", "3"])])
def test_generated_polytope_renders_like_the_references(name, symbol):
    g = fx.load(name)
    n = int(g["dimension"])
    p = polytope.RegularPolytope(symbol)
    assert abs(-4 * p.circumradius() - float(g["cam_distance"])) < 2e-6 * abs(float(g["cam_distance"]))
    boundary, root = tracern.build_kdtree(p.hull(NTracer(n), MAT, max_edge=0))
    flat = tracern.CompositeScene._flatten(boundary, root)
    flat["batch_size"] = 4
    w, h = int(g["width"]), int(g["height"])
    for k in (0, 1):
        f = g["frames"][k]
        c = ob.OracleScene(n, g["origins"][f], g["axes"][f], flat=flat).colors_at(g["xs"], g["ys"], w, h)
        d = np.abs(c - g["colors"][k]).max(axis=1)
        assert (d > 1e-3).sum() <= 3, (name, int((d > 1e-3).sum()))
        assert np.median(d) < 1e-6


@pytest.mark.parametrize("name,n", [("box_n6_1920x1080", 6), ("cell120_n4", 4)])
def test_rotating_cameras_match_the_captured_sequence(name, n):
    g = fx.load(name)
    nt = NTracer(n)
    dist = -math.sqrt(n) * 4 if name.startswith("box") else float(g["cam_distance"])
    for f, cam in enumerate(polytope.rotating_cameras(nt, dist, 160)):
        if f >= 30:
            break
        assert np.abs(np.array(list(cam.origin)) - g["origins"][f]).max() < 2e-5, f
        assert np.abs(np.array([list(cam.axes[i]) for i in range(n)]) - g["axes"][f]).max() < 2e-6, f

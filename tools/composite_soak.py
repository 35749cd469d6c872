#!/usr/bin/env python3
"""Composite-scene soak (development aid): the golden scenes under random cameras, every frame rendered through the drop-in call
(BlockingRenderer.render, three fp32 channels) and compared with the oracle's frame, colour by colour.
    python3 tools/composite_soak.py [frames_per_scene [seed [WxH [scenes]]]]
Variants per scene: as captured; "rebuilt" -- the lit scene under the native builder's k-d tree; "lit" -- a point light outside and one inside the scene's box, a global light, shadows on,
nothing reflective (the per-lane shadow walk with its far-child rule); "mirror" -- every material 30 % reflective, depth 2.
Prints one line per (scene, variant): frames, worst |difference|, pixels beyond 1e-5."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ntracer_amd  # noqa: E402
from ntracer_amd import tracern  # noqa: E402
import fixtures as fx  # noqa: E402
import oracle_binding as ob  # noqa: E402

TOL = 1e-5
RGBF32 = fx.RGBF32


def cameras(rng, n, dist, count):
    out = []
    for k in range(count):
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        q = np.ascontiguousarray(q, np.float32)
        d = np.float32(dist * rng.choice([0.35, 0.6, 1.0, 1.0, 1.5, 2.5]))
        o = -q[2] * d + np.float32(rng.uniform(-0.2, 0.2) * d) * q[0] + np.float32(rng.uniform(-0.2, 0.2) * d) * q[1]
        out.append((o.astype(np.float32), q))
    return out


def soak_scene(name, per, seed, W, H, threads):
    """The three variants of one golden scene under `per` random cameras each: [(variant, n, worst |diff|, values beyond TOL,
    frames with such values, primary hits, shadow rays)]."""
    fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in RGBF32])
    g = fx.load(name)
    n = int(g["dimension"])
    base = fx.params_of(g)
    dist = abs(float(g["cam_distance"])) if "cam_distance" in g.files else float(np.linalg.norm(g["origins"][0]))
    lo, hi = np.asarray(g["aabb_start"], np.float32), np.asarray(g["aabb_end"], np.float32)
    ctr, ext = 0.5 * (lo + hi), 0.5 * (hi - lo)
    variants = [("captured", fx.flat_of(g), dict(base))]
    lit = dict(base)
    out_pos = ctr + ext * 3.0 * np.resize(np.array([1.0, 0.8, -0.9, 0.4], np.float32), n)
    in_pos = ctr + ext * 0.15 * np.resize(np.array([-0.5, 0.3, 0.2, -0.4], np.float32), n)
    gdir = np.resize(np.array([0.2, -0.9, 0.3, 0.1], np.float32), n)
    lit.update(shadows=1, point_light_pos=[list(map(float, out_pos)), list(map(float, in_pos))],
               point_light_color=[[float(40.0 * np.linalg.norm(ext) ** (n - 1))] * 3, [float(0.5 * np.linalg.norm(ext) ** (n - 1))] * 3],
               global_light_dir=[list(map(float, gdir / np.linalg.norm(gdir)))], global_light_color=[[0.4, 0.4, 0.5]], ambient=[0.02, 0.02, 0.03])
    flat_plain = fx.flat_of(g)
    m = np.array(flat_plain["materials"], np.float32).copy()
    m[:, 7] = 0.0
    flat_plain["materials"] = m
    variants.append(("lit", flat_plain, lit))
    flat_mirror = fx.flat_of(g)
    m = np.array(flat_mirror["materials"], np.float32).copy()
    m[:, 7] = 0.3
    flat_mirror["materials"] = m
    mir = dict(lit)
    mir.update(max_reflect_depth=2)
    variants.append(("mirror", flat_mirror, mir))
    # the same primitives, lit, under the k-d tree of the native builder (with shadows on the pixels depend on the tree, here as
    # in the reference: the oracle walks the same rebuilt tree)
    reb = tracern.CompositeScene.from_flat(n, flat_plain).with_rebuilt_tree()
    variants.append(("rebuilt", dict(reb._flat), lit))
    out = []
    for vname, flat, params in variants:
        rng = np.random.default_rng(seed + n * 131 + len(vname))
        cams = cameras(rng, n, dist, per)
        sc = tracern.CompositeScene.from_flat(n, flat)
        sc.set_params_flat(params)
        osc = ob.OracleScene(n, cams[0][0], cams[0][1], flat=flat, params=params)
        worst, nbad, frames_bad, hits, shadow = 0.0, 0, 0, 0, 0
        for o, a in cams:
            sc._set_camera_arrays(o, a)
            buf = bytearray(fmt.pitch * H)
            assert ntracer_amd.BlockingRenderer().render(buf, fmt, sc)
            got = np.frombuffer(bytes(buf), np.uint8).reshape(H, fmt.pitch).view(">f4")
            osc.set_camera(o, a)
            ref, cnt = osc.render(W, H, RGBF32, threads=threads, counters=True)
            ref = ref.view(">f4")
            hits += cnt.get("hits", 0)
            shadow += cnt.get("shadow_rays", 0)
            d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
            worst = max(worst, float(d.max()))
            k = int((d > TOL).sum())
            nbad += k
            frames_bad += 1 if k else 0
        out.append((vname, n, worst, nbad, frames_bad, hits, shadow))
    return out


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4100
    W, H = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "320x200").split("x"))
    names = sys.argv[4].split(",") if len(sys.argv) > 4 else ["cell120_n4", "cell600_n4", "orthoplex5_n5", "simplex7_n7", "simplex9_n9", "simplex10_n10",
                                                              "feature5_n5", "feature11_n11", "lit12_n12", "feature16_n16"]
    import bench
    threads = max(1, min(64, bench.cpu_quota_cores() - 1))
    bad_total = 0
    for name in names:
        for vname, n, worst, nbad, frames_bad, hits, shadow in soak_scene(name, per, seed, W, H, threads):
            print("%-14s %-8s n=%-2d %3d frames %dx%d (%4.1f %% of the primary rays hit, %d shadow rays): worst |diff| %.2e, %d values beyond %.0e in %d frames"
                  % (name, vname, n, per, W, H, 100.0 * hits / (per * W * H), shadow, worst, nbad, TOL, frames_bad), flush=True)
            bad_total += nbad
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Quick GPU-vs-oracle diagnostics (development aid; the real parity tests are tests/ -m gpu).
Prints max |delta| and mismatch counts per scene, and simple timings."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ntracer_amd  # noqa: E402
from ntracer_amd import tracern  # noqa: E402
import oracle_binding as ob  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
RGBX = [(8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1), (8, 0, 0, 0)]
PK = ["fov", "shadows", "camera_light", "max_reflect_depth", "bg_gradient_axis", "ambient", "bg1", "bg2", "bg3",
      "point_light_pos", "point_light_color", "global_light_dir", "global_light_color"]


def fmt_of(w, h, chans, pitch=0, rev=False):
    return ntracer_amd.ImageFormat(w, h, [ntracer_amd.Channel(*c) for c in chans], pitch, rev)


def box():
    for name in ["box_n3_1920x1080", "box_n6_1920x1080", "box_n10_4096x4096", "box_n5_320x200", "box_n8_320x200",
                 "box_n12_320x200"]:
        g = np.load(os.path.join(G, name + ".npz"))
        n = g["origins"].shape[1]
        w, h = int(g["width"]), int(g["height"])
        sc = tracern.BoxScene(n)
        worst = 0.0
        exact = 0
        tot = 0
        worst_ref = 0.0
        bad_ref = 0
        for k, f in enumerate(g["frames"]):
            sc._set_camera_arrays(g["origins"][f], g["axes"][f])
            c = sc.colors_at(g["xs"], g["ys"], w, h)
            o = ob.OracleScene(n, g["origins"][f], g["axes"][f], float(g["fov"])).colors_at(g["xs"], g["ys"], w, h)
            d = np.abs(c - o)
            worst = max(worst, float(d.max()))
            exact += int((d.max(axis=1) == 0).sum())
            tot += len(d)
            dr = np.abs(c - g["colors"][k]).max(axis=1)
            worst_ref = max(worst_ref, float(dr.max()))
            bad_ref += int((dr > 1e-4).sum())
        print("%-22s gpu-vs-oracle max %.3g exact %d/%d | gpu-vs-reference max %.3g bad %d" % (name, worst, exact, tot, worst_ref, bad_ref))
        # full-frame bytes vs oracle
        sc._set_camera_arrays(g["origins"][17], g["axes"][17])
        W, H = (480, 270)
        fmt = fmt_of(W, H, RGBX)
        buf = bytearray(fmt.pitch * H)
        ntracer_amd.BlockingRenderer().render(buf, fmt, sc)
        ref = ob.OracleScene(n, g["origins"][17], g["axes"][17], float(g["fov"])).render(W, H, RGBX, threads=7)
        got = np.frombuffer(bytes(buf), np.uint8).reshape(H, fmt.pitch)
        print("    480x270 RGBX8 bytes differing:", int((got != ref).sum()))


def packing():
    g = np.load(os.path.join(G, "packing_box3.npz"))
    w, h = int(g["width"]), int(g["height"])
    sc = tracern.BoxScene(3)
    sc._set_camera_arrays(g["origin"], g["axes"])
    for name in g["names"]:
        tab = g["fmt_%s_channels" % name]
        pitch, rev, bpp = [int(v) for v in g["fmt_%s_meta" % name]]
        chans = ob.channels_from_table(tab)
        fmt = fmt_of(w, h, chans, pitch, bool(rev))
        buf = bytearray(pitch * h)
        ntracer_amd.BlockingRenderer().render(buf, fmt, sc)
        got = np.frombuffer(bytes(buf), np.uint8).reshape(h, pitch)[:, :w * bpp]
        ref = g["fmt_%s_image" % name][:, :w * bpp]
        print("packing %-18s bpp %2d differing bytes %d" % (name, bpp, int((got != ref).sum())))


def composite():
    for name in ["cell600_n4", "cell120_n4"]:
        g = np.load(os.path.join(G, name + ".npz"))
        n = int(g["dimension"])
        w, h = int(g["width"]), int(g["height"])
        t0 = time.time()
        sc = tracern.CompositeScene.from_flat(n, g)
        flat = {k: g[k] for k in tracern._FLAT_KEYS}
        flat["batch_size"] = 4
        print(name, "scene create %.2fs" % (time.time() - t0))
        for k, f in enumerate(g["frames"]):
            sc._set_camera_arrays(g["origins"][f], g["axes"][f])
            c = sc.colors_at(g["xs"], g["ys"], w, h)
            o = ob.OracleScene(n, g["origins"][f], g["axes"][f], flat=flat).colors_at(g["xs"], g["ys"], w, h)
            d = np.abs(c - o).max(axis=1)
            dr = np.abs(c - g["colors"][k]).max(axis=1)
            print("   frame %3d gpu-vs-oracle max %.3g exact %d/%d >1e-4: %d | vs reference max %.3g >1e-4: %d" %
                  (f, d.max(), int((d == 0).sum()), len(d), int((d > 1e-4).sum()), dr.max(), int((dr > 1e-4).sum())))
        # full frame timing + stats
        W, H = 1920, 1080
        fmt = fmt_of(W, H, RGBX)
        buf = bytearray(fmt.pitch * H)
        sc._set_camera_arrays(g["origins"][0], g["axes"][0])
        r = ntracer_amd.BlockingRenderer()
        r.render(buf, fmt, sc)
        t0 = time.time()
        r.render(buf, fmt, sc)
        dt = time.time() - t0
        print("   1080p host render %.2f ms (%.1f Mrays/s incl. D2H)" % (dt * 1e3, W * H / dt / 1e6))
        r.render(buf, fmt, sc, collect_stats=True)
        st = sc.last_stats()
        print("   stats/ray:", {k: round(v / max(st["rays"], 1), 2) for k, v in st.items()})
        got = np.frombuffer(bytes(buf), np.uint8).reshape(H, fmt.pitch)
        if name == "cell600_n4":
            ref = ob.OracleScene(n, g["origins"][0], g["axes"][0], flat=flat).render(W, H, RGBX, threads=7)
            print("   1080p bytes differing vs oracle:", int((got != ref).sum()), "max", int(np.abs(got.astype(int) - ref.astype(int)).max()))


def feature():
    g = np.load(os.path.join(G, "feature3d.npz"))
    flat = {k: g[k] for k in tracern._FLAT_KEYS}
    flat["batch_size"] = 4
    # opaque variant of the feature scene: force every material opaque so the GPU path accepts it
    flat_op = dict(flat)
    mats = g["materials"].copy()
    mats[:, 6] = 1.0
    flat_op["materials"] = mats
    w, h = int(g["width"]), int(g["height"])
    ys, xs = np.mgrid[0:h, 0:w]
    xs = xs.ravel()
    ys = ys.ravel()
    sc = tracern.CompositeScene.from_flat(3, flat_op)
    sc._set_camera_arrays(g["origin"], g["axes"])
    for v in g["variants"]:
        params = {k: g["%s__%s" % (v, k)] for k in PK}
        sc.set_params_flat(params)
        c = sc.colors_at(xs, ys, w, h)
        o = ob.OracleScene(3, g["origin"], g["axes"], flat=flat_op, params=params, clean_normals=True).colors_at(xs, ys, w, h)
        os_ = ob.OracleScene(3, g["origin"], g["axes"], flat=flat_op, params=params).colors_at(xs, ys, w, h)
        print("      oracle clean-vs-strict pixels differing:", int((np.abs(o - os_).max(axis=1) > 1e-4).sum()))
        d = np.abs(c - o).max(axis=1)
        print("feature3d(opaque) %-14s gpu-vs-oracle max %.3g  >1e-4: %d / %d  >1e-5: %d" % (v, d.max(), int((d > 1e-4).sum()), len(d), int((d > 1e-5).sum())))
        for i in np.nonzero(d > 1e-4)[0][:4]:
            print("      px", xs[i], ys[i], "gpu", c[i], "oracle", o[i])


def box_timing():
    import torch
    for n, (W, H) in ((3, (1920, 1080)), (6, (1920, 1080)), (10, (4096, 4096))):
        sc = tracern.BoxScene(n)
        g = np.load(os.path.join(G, "box_n%d_%dx%d.npz" % (n, W, H)))
        sc._set_camera_arrays(g["origins"][17], g["axes"][17])
        for chans, label in ((RGBX, "rgbx8"), ([(32, 1, 0, 0, 0, True), (32, 0, 1, 0, 0, True), (32, 0, 0, 1, 0, True)], "f32x3")):
            fmt = fmt_of(W, H, chans)
            fb = torch.empty(fmt.pitch * H, dtype=torch.uint8, device="cuda")
            r = ntracer_amd.BlockingRenderer()
            for _ in range(5):
                r.render(fb, fmt, sc)
            torch.cuda.synchronize()
            K = 50
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(K):
                r.render(fb, fmt, sc)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / K
            print("box n=%d %dx%d %s: %.1f us/frame, %.1f Grays/s, %.2f TB/s framebuffer" % (n, W, H, label, ms * 1e3, W * H / ms / 1e6, fmt.pitch * H / ms / 1e9))


if __name__ == "__main__":
    which = sys.argv[1:] or ["box", "packing", "composite", "feature", "box_timing"]
    for wname in which:
        print("=====", wname)
        globals()[wname]()

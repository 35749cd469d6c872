"""Parity tests proper: the HIP path, called through the C ABI, against
  (1) the oracle on the same inputs,
  (2) the golden vectors captured from the compiled reference,
  (3) size-independent properties at BASELINE.json's full sizes.

Tolerances: colours are fp32; north_star demands max per-channel |delta| < 1e-4.  Because the kernels
mirror the oracle's operation order (no FMA contraction, IEEE div/sqrt) the observed difference is 0 for
BoxScene and <= 2.4e-7 (powf) for composite scenes; the tests assert 1e-5 against the oracle and 1e-4
against the reference (whose -ffast-math build differs in the last bits).  Packed bytes are compared
exactly.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import fixtures as fx
import ntracer_amd
import oracle_binding as ob
from ntracer_amd import _lib, tracern
from ntracer_amd import distributed as ntd

pytestmark = pytest.mark.gpu

TOL_ORACLE = 1e-5
TOL_REF = 1e-4


def fmt_of(w, h, chans, pitch=0, rev=False):
    return ntracer_amd.ImageFormat(w, h, [ntracer_amd.Channel(*c) for c in chans], pitch, rev)


def render_host(scene, fmt, **kw):
    buf = bytearray(fmt.pitch * fmt.height)
    assert ntracer_amd.BlockingRenderer().render(buf, fmt, scene, **kw)
    return np.frombuffer(bytes(buf), np.uint8).reshape(fmt.height, fmt.pitch)


def test_extension_is_loaded_and_sees_the_gpu():
    assert _lib.lib().nt_device_count() >= 1
    loaded = open("/proc/self/maps").read()
    assert "libntracer_hip.so" in loaded


# ------------------------------------------------------------------ BoxScene (configs 1, 2, 3, 5)
@pytest.mark.parametrize("name", fx.BOX_FIXTURES)
def test_box_bit_exact_vs_oracle_and_within_tol_of_reference(name):
    g = fx.load(name)
    n = g["origins"].shape[1]
    w, h = int(g["width"]), int(g["height"])
    sc = tracern.BoxScene(n)
    bad_ref = 0
    total = 0
    for k, f in enumerate(g["frames"]):
        sc._set_camera_arrays(g["origins"][f], g["axes"][f])
        c = sc.colors_at(g["xs"], g["ys"], w, h)
        o = ob.OracleScene(n, g["origins"][f], g["axes"][f], float(g["fov"])).colors_at(g["xs"], g["ys"], w, h)
        assert np.array_equal(c, o), (name, int(f))                    # integer-exact fp32 agreement
        bad_ref += int((np.abs(c - g["colors"][k]).max(axis=1) > TOL_REF).sum())
        total += len(c)
    assert bad_ref <= max(1, total // 20000)


def test_config1_box3_256_bytes_equal_reference_image():
    g = fx.load("box_cfg1_n3_256")
    sc = tracern.BoxScene(3)
    sc._set_camera_arrays(g["origin"], g["axes"])
    img = render_host(sc, fmt_of(256, 256, fx.RGBX8))
    assert np.array_equal(img, g["image_rgbx8"])
    c = sc.calculate_color(128, 128, 256, 256)
    o = ob.OracleScene(3, g["origin"], g["axes"]).colors_at([128], [128], 256, 256)[0]
    assert tuple(c) == tuple(float(v) for v in o)


@pytest.mark.parametrize("n,w,h", [(3, 1920, 1080), (6, 1920, 1080)])
def test_box_full_frame_bytes_equal_oracle(n, w, h):
    g = fx.load("box_n%d_%dx%d" % (n, w, h))
    sc = tracern.BoxScene(n)
    for f in (0, 93):
        sc._set_camera_arrays(g["origins"][f], g["axes"][f])
        img = render_host(sc, fmt_of(w, h, fx.RGBX8))
        ref = ob.OracleScene(n, g["origins"][f], g["axes"][f]).render(w, h, fx.RGBX8, threads=7)
        assert np.array_equal(img, ref)


def _stress_cameras(n, rng):
    """orientations x origins chosen to land rays on edges, faces' planes, the inside, grazing directions"""
    cams = []
    eye = np.eye(n, dtype=np.float32)
    for k in range(10):
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        if k == 0:
            q = eye.copy()                                   # axis-aligned: direction components exactly 0
        elif k == 1:
            q = eye + 1e-7 * rng.standard_normal((n, n))     # almost axis-aligned: grazing rays
        elif k == 2:
            q = eye[rng.permutation(n)]
        elif k == 3:
            q = q.copy()
            q[1] = q[1] + 0.8 * q[2] + 0.3 * q[0]            # `up` far from orthogonal: no quadratic |dir|^2 shortcut
        elif k == 4:
            q = q.copy()
            q[0] = 3.0 * q[0]                                # stretched `right`
            q[1] = 0.0 * q[1]                                # ... and no `up` at all: every row the same
        q = np.ascontiguousarray(q, np.float32)
        for dist in (0.3, 1.0, 1.0000001, 1.7, 3.0, 9.0, 60.0):
            back = -q[2] * np.float32(dist)                  # look at the centre from `dist` away ...
            cams.append((back.astype(np.float32), q))
            off = back + np.float32(0.4) * q[0] + np.float32(0.25) * q[1]          # ... and off-centre
            cams.append((off.astype(np.float32), q))
        o = np.zeros(n, np.float32)
        o[:3] = (-1.0, 0.3, -2.5)                            # origin exactly on the plane of a face
        cams.append((o, q))
        o = o.copy()
        o[0] = 1.0
        o[min(3, n - 1)] = 1.0                               # on two planes at once
        cams.append((o, q))
    return cams


@pytest.mark.parametrize("n", [3, 4, 6, 8, 10, 11, 13, 15, 16, 20, 21, 24, 27, 40, 64])
def test_box_stress_cameras_bytes_and_floats_equal_oracle(n):
    """The BoxScene kernel sorts rays into clear misses, clear hits and unclear ones, and only the last get the
    reference-ordered evaluation; quantised formats also skip the sqrt and the division away from rounding
    boundaries.  Every shortcut is guarded, so whole frames must still equal the oracle byte for byte: RGBX8
    (the specialised kernel), RGB565-like 5-bit channels, 16-bit channels and fp32 channels (the general one)."""
    rng = np.random.default_rng(1234 + n)
    w, h = 320, 72
    formats = [fx.RGBX8,
               [(5, 1, 0, 0), (6, 0, 1, 0), (5, 0, 0, 1)],
               [(16, 0, 0, 1), (16, 0, 1, 0), (16, 1, 0, 0)],
               [(32, 1, 0, 0, 0, True), (32, 0, 1, 0, 0, True), (32, 0, 0, 1, 0, True)]]
    sc = tracern.BoxScene(n)
    for k, (origin, axes) in enumerate(_stress_cameras(n, rng)):
        sc._set_camera_arrays(origin, axes)
        osc = ob.OracleScene(n, origin, axes)
        chans = formats[k % len(formats)] if k % 3 else fx.RGBX8
        img = render_host(sc, fmt_of(w, h, chans))
        ref = osc.render(w, h, chans, threads=7)
        assert np.array_equal(img, ref), (n, k, int((img != ref).sum()))
        if k % 5 == 0:
            xs = rng.integers(0, w, 2000)
            ys = rng.integers(0, h, 2000)
            got = sc.colors_at(xs, ys, w, h)
            assert np.array_equal(got.view(np.uint32), osc.colors_at(xs, ys, w, h).view(np.uint32)), (n, k)


@pytest.mark.parametrize("rev", [False, True])
def test_box_byte_orders_of_packed_rgb(rev):
    """RGBX / BGRX / XRGB / XBGR / RGB with a repeated component, straight and byte-reversed: the byte-permute packing
    of the specialised BoxScene kernel against the oracle's generic packer."""
    layouts = [[(8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1), (8, 0, 0, 0)],
               [(8, 0, 0, 1), (8, 0, 1, 0), (8, 1, 0, 0), (8, 0, 0, 0)],
               [(8, 0, 0, 0), (8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1)],
               [(8, 0, 0, 0), (8, 0, 0, 1), (8, 0, 1, 0), (8, 1, 0, 0)],
               [(8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1), (8, 1, 0, 0)],
               [(8, 0, 1, 0), (8, 0, 0, 0), (8, 0, 0, 0), (8, 0, 1, 0)]]
    g = fx.load("box_n6_1920x1080")
    w, h = 448, 40
    sc = tracern.BoxScene(6)
    for k, chans in enumerate(layouts):
        f = 20 * k + 3
        sc._set_camera_arrays(g["origins"][f], g["axes"][f])
        img = render_host(sc, fmt_of(w, h, chans, 0, rev))
        ref = ob.OracleScene(6, g["origins"][f], g["axes"][f]).render(w, h, chans, reversed_=rev, threads=7)
        assert np.array_equal(img, ref), (k, rev, int((img != ref).sum()))


def test_run_time_n_rows_kernel_on_the_bench_frames(monkeypatch):
    """NTRACER_FORCE_VAR=1 sends BoxScene(6) through box_rows_kernel_var (stretch codes and lean loops with run-time n): a
    multi-frame launch of bench cameras at 1920x1080 and a 10-D frame at 4096 columns, byte for byte against the frames
    of the compile-time-N kernels (which other tests pin to the oracle); and the plain per-pixel kernel
    (NTRACER_BOX_VAR_ROWS=0) gives the same."""
    import torch
    g = fx.load("box_n6_1920x1080")
    w, h = 1920, 1080
    fmt = fmt_of(w, h, fx.RGBX8)
    frames = [0, 17, 40, 77, 93, 120, 141, 159]
    o = np.ascontiguousarray(g["origins"][frames], np.float32)
    a = np.ascontiguousarray(g["axes"][frames], np.float32)
    st_ = fmt._as_struct()

    def launch():
        sc = tracern.BoxScene(6)
        fb = torch.zeros((len(frames), h * fmt.pitch), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), h * fmt.pitch, len(frames), o.ctypes.data_as(_lib.f32p),
                                                      a.ctypes.data_as(_lib.f32p), C.byref(st_), None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        return fb.cpu().numpy()

    fixed = launch()
    monkeypatch.setenv("NTRACER_FORCE_VAR", "1")
    rows = launch()
    assert np.array_equal(rows, fixed), int((rows != fixed).sum())
    monkeypatch.setenv("NTRACER_BOX_VAR_ROWS", "0")
    plain = launch()
    assert np.array_equal(plain, fixed)


def test_box_wide_rows_and_the_kernel_without_stretch_codes(monkeypatch):
    """More than 2048 pixels a row (two words of redo bits, several of stretch codes), a width that is not a multiple
    of 64, and the same frames through the general kernel (NTRACER_BOX_CULL=0: no pre-kernel, no second pass)."""
    rng = np.random.default_rng(77)
    for n, w, h in ((4, 4100, 11), (6, 2307, 19), (8, 130, 70)):
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        axes = np.ascontiguousarray(q, np.float32)
        origin = (-axes[2] * np.float32(2.2) + np.float32(0.3) * axes[0]).astype(np.float32)
        sc = tracern.BoxScene(n)
        sc._set_camera_arrays(origin, axes)
        ref = ob.OracleScene(n, origin, axes).render(w, h, fx.RGBX8, threads=7)
        img = render_host(sc, fmt_of(w, h, fx.RGBX8))
        assert np.array_equal(img, ref), (n, w, h, int((img != ref).sum()))
        monkeypatch.setenv("NTRACER_BOX_CULL", "0")
        img0 = render_host(sc, fmt_of(w, h, fx.RGBX8))
        monkeypatch.delenv("NTRACER_BOX_CULL")
        assert np.array_equal(img0, ref), (n, w, h)


def test_box10_4096_both_kernels_properties_and_samples(monkeypatch):
    """config 5 at full size, 4096x4096, through the compile-time-N kernel (default for n <= 10) and through the
    run-time-n kernel (NTRACER_FORCE_VAR; the only one for n > 10): identical frames.  Oracle on sampled rows; at
    full size the frame must be invariant under the band decomposition (2 ranks, compact) and identical between
    the host path and the device path."""
    import torch
    g = fx.load("box_n10_4096x4096")
    w = h = 4096
    sc = tracern.BoxScene(10)
    sc._set_camera_arrays(g["origins"][40], g["axes"][40])
    fmt = fmt_of(w, h, fx.RGBX8)
    img = render_host(sc, fmt)
    monkeypatch.setenv("NTRACER_FORCE_VAR", "1")
    assert np.array_equal(render_host(sc, fmt), img)
    monkeypatch.delenv("NTRACER_FORCE_VAR")
    osc = ob.OracleScene(10, g["origins"][40], g["axes"][40])
    for y in (0, 1777, 2048, 4095):
        xs = np.arange(w)
        packed = np.concatenate([np.frombuffer(ob.pack_pixel(c, fx.RGBX8), np.uint8)
                                 for c in osc.colors_at(xs, np.full(w, y), w, h)])
        assert np.array_equal(img[y], packed), y
    # band decomposition invariance
    parts = []
    for r in range(2):
        rows = ntd.owned_rows(h, r, 2)
        buf = bytearray(len(rows) * fmt.pitch)
        assert ntracer_amd.BlockingRenderer().render(buf, fmt, sc, band_rank=r, band_world=2, compact=True)
        parts.append((rows, np.frombuffer(bytes(buf), np.uint8).reshape(len(rows), fmt.pitch)))
    re = np.empty_like(img)
    for rows, p in parts:
        re[rows] = p
    assert np.array_equal(re, img)
    # device-resident render == host render
    fb = torch.zeros(fmt.pitch * h, dtype=torch.uint8, device="cuda")
    assert ntracer_amd.BlockingRenderer().render(fb, fmt, sc)
    torch.cuda.synchronize()
    assert np.array_equal(fb.cpu().numpy().reshape(h, fmt.pitch), img)


def test_pixel_packing_all_formats_bit_exact_vs_reference():
    g = fx.load("packing_box3")
    w, h = int(g["width"]), int(g["height"])
    sc = tracern.BoxScene(3)
    sc._set_camera_arrays(g["origin"], g["axes"])
    for name in g["names"]:
        pitch, rev, bpp = [int(v) for v in g["fmt_%s_meta" % name]]
        chans = ob.channels_from_table(g["fmt_%s_channels" % name])
        buf = bytearray(b"\xAB" * (pitch * h))
        assert ntracer_amd.BlockingRenderer().render(buf, fmt_of(w, h, chans, pitch, bool(rev)), sc)
        got = np.frombuffer(bytes(buf), np.uint8).reshape(h, pitch)
        assert np.array_equal(got[:, :w * bpp], g["fmt_%s_image" % name][:, :w * bpp]), str(name)
        assert (got[:, w * bpp:] == 0xAB).all(), "pitch padding must not be touched: %s" % name


def test_random_channel_layouts_match_oracle_bytes():
    """Seeded random formats: 1..9 channels, 1..31-bit integers and 32-bit floats, negative / > 1 weights, up to
    128 bits, reversed or not, padded pitch -- through every packing path (32-bit, 64-bit, generic 128-bit)."""
    rnd = np.random.RandomState(20260101)
    g = fx.load("box_n5_320x200")
    sc = tracern.BoxScene(5)
    sc._set_camera_arrays(g["origins"][40], g["axes"][40])
    osc = ob.OracleScene(5, g["origins"][40], g["axes"][40])
    w, h = 53, 31
    seen_modes = set()
    for trial in range(48):
        chans = []
        bits = 0
        for _ in range(rnd.randint(1, 10)):
            if rnd.rand() < 0.2 and bits + 32 <= 128:
                chans.append((32, float(rnd.uniform(-1, 2)), float(rnd.uniform(-1, 2)), float(rnd.uniform(-1, 2)), float(rnd.uniform(-.5, .5)), True))
                bits += 32
            else:
                b = int(rnd.choice([1, 2, 5, 8, 10, 16, 24, 29, 30, 31]))
                if bits + b > 128:
                    break
                if rnd.rand() < 0.25:
                    chans.append((b, 0, 0, 0))                       # padding channel
                else:
                    chans.append((b, float(rnd.uniform(-1, 2)), float(rnd.uniform(-1, 2)), float(rnd.uniform(-1, 2)), float(rnd.uniform(-.5, .5))))
                bits += b
        if not chans:
            continue
        bpp = (bits + 7) // 8
        pitch = w * bpp + int(rnd.choice([0, 0, 3, 8]))
        rev = bool(rnd.rand() < 0.5)
        live = sum(1 for c in chans if any(c[1:5]) or (len(c) > 5 and c[5]))
        seen_modes.add("w32" if bits <= 32 and live <= 4 else "w64" if bits <= 64 and live <= 4 else "gen")
        img = render_host(sc, fmt_of(w, h, chans, pitch, rev))
        ref = osc.render(w, h, chans, pitch, rev)
        assert np.array_equal(img[:, :w * bpp], ref[:, :w * bpp]), (trial, chans, rev)
    assert seen_modes == {"w32", "w64", "gen"}


def test_ragged_and_tiny_images():
    g = fx.load("box_n6_1920x1080")
    sc = tracern.BoxScene(6)
    sc._set_camera_arrays(g["origins"][17], g["axes"][17])
    osc = ob.OracleScene(6, g["origins"][17], g["axes"][17])
    for (w, h) in [(1, 1), (1, 7), (63, 5), (65, 3), (257, 33), (31, 129)]:
        for chans in (fx.RGBX8, fx.RGB16, fx.RGBF32):
            img = render_host(sc, fmt_of(w, h, chans))
            assert np.array_equal(img, osc.render(w, h, chans)), (w, h)


def test_row_tables_of_many_launch_geometries():
    """The tile kernel reads sy and the row offsets from a table the host keeps per launch geometry (view, fov, pitch, band
    split; a handful cached, the oldest dropped): a dozen geometries on one scene, the first ones again afterwards, a
    changed fov, padded pitches and band splits -- every image equals the oracle's."""
    g = fx.load("box_n6_1920x1080")
    sc = tracern.BoxScene(6)
    sc._set_camera_arrays(g["origins"][23], g["axes"][23])
    osc = ob.OracleScene(6, g["origins"][23], g["axes"][23])
    sizes = [(200, 40 + 3 * k) for k in range(12)] + [(200, 40), (200, 43)]
    for (w, h) in sizes:
        for pitch in (0, 4 * w + 64):
            img = render_host(sc, fmt_of(w, h, fx.RGBX8, pitch))
            ref = osc.render(w, h, fx.RGBX8)
            assert np.array_equal(img[:, :4 * w], ref), (w, h, pitch)
    sc.set_fov(1.1)
    osc2 = ob.OracleScene(6, g["origins"][23], g["axes"][23], 1.1)
    assert np.array_equal(render_host(sc, fmt_of(200, 40, fx.RGBX8)), osc2.render(200, 40, fx.RGBX8))
    # rows dealt in bands, into a full-size frame (rows of a wave are not `pitch` apart) and into a compact one
    import torch
    w, h = 320, 200
    fmt = fmt_of(w, h, fx.RGBX8)
    full = osc2.render(w, h, fx.RGBX8)
    fst = fmt._as_struct()
    o1 = np.ascontiguousarray(g["origins"][23:24], np.float32)
    a1 = np.ascontiguousarray(g["axes"][23:24], np.float32)
    from ntracer_amd import distributed as ntd
    for world, rows in ((3, 8), (4, 16)):
        for compact in (0, 1):
            for rank in range(world):
                own = ntd.owned_rows(h, rank, world, rows)
                opts = _lib.NtRenderOpts()
                opts.device = 0
                opts.band_rank, opts.band_world, opts.band_rows, opts.compact = rank, world, rows, compact
                nrows = len(own) if compact else h
                fb = torch.zeros((nrows * fmt.pitch,), dtype=torch.uint8, device="cuda")
                _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), nrows * fmt.pitch, 1, o1.ctypes.data_as(_lib.f32p),
                                                              a1.ctypes.data_as(_lib.f32p), C.byref(fst), C.byref(opts),
                                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                torch.cuda.synchronize()
                got = fb.cpu().numpy().reshape(nrows, fmt.pitch)
                if compact:
                    assert np.array_equal(got, full[own]), (world, rows, rank)
                else:
                    assert np.array_equal(got[own], full[own]), (world, rows, rank)


# ------------------------------------------------------------------ CompositeScene (config 4)
def test_simplex10_fixed_and_run_time_n_kernels_agree(monkeypatch):
    """A 10-D composite scene renders through launch_composite_fixed<10> by default and through
    composite_kernel_var when forced: same colours (both mirror the oracle's operation order)."""
    g = fx.load("simplex10_n10")
    sc = tracern.CompositeScene.from_flat(10, fx.flat_of(g))
    f = g["frames"][2]
    sc._set_camera_arrays(g["origins"][f], g["axes"][f])
    fmt = fmt_of(320, 200, fx.RGBF32)
    a = render_host(sc, fmt)
    monkeypatch.setenv("NTRACER_FORCE_VAR", "1")
    b = render_host(sc, fmt)
    assert np.abs(a.view(">f4") - b.view(">f4")).max() < 1e-6
    assert a.view(">f4").max() > 0.2


@pytest.mark.parametrize("force_var", [False, True])
def test_lit_reflective_scene_in_ten_dimensions_vs_oracle(monkeypatch, force_var):
    """n = 9 and 10 go through the compile-time-N kernels, i.e. with the full feature set (the reference's generic
    module has it for any n): lights, shadows and reflection on the 10-D simplex against the oracle.  force_var: the same
    scene through the run-time-n kernel (composite_kernel_var), which has the same feature set."""
    if force_var:
        monkeypatch.setenv("NTRACER_FORCE_VAR", "1")
    g = fx.load("simplex10_n10")
    flat = fx.flat_of(g)
    m = np.array(flat["materials"], np.float32).copy()
    m[:, 7] = 0.25
    flat["materials"] = m
    n = 10
    params = dict(fov=0.8, shadows=1, camera_light=1, max_reflect_depth=2, bg_gradient_axis=1,
                  ambient=[.02, .02, .03], bg1=[1, 1, 1], bg2=[0, 0, 0], bg3=[0, 1, 1],
                  point_light_pos=[[6.0, 5.0, -7.0, 2.0, 1.0, -1.0, 0.5, 0.0, 2.0, -3.0]], point_light_color=[[9e8, 8e8, 7e8]],
                  global_light_dir=[[0.2, -0.9, 0.3, 0.1, 0.0, 0.1, -0.1, 0.0, 0.05, 0.0]], global_light_color=[[.4, .4, .5]])
    sc = tracern.CompositeScene.from_flat(n, flat)
    sc.set_params_flat(params)
    f = g["frames"][1]
    sc._set_camera_arrays(g["origins"][f], g["axes"][f])
    fmt = fmt_of(160, 100, fx.RGBF32)
    img = render_host(sc, fmt)
    ref = ob.OracleScene(n, g["origins"][f], g["axes"][f], flat=flat, params=params, clean_normals=True).render(160, 100, fx.RGBF32, threads=7)
    d = np.abs(img.view(">f4") - ref.view(">f4"))
    assert d.max() < 1e-4, float(d.max())
    assert img.view(">f4").max() > 0.3


def test_twelve_dimensional_lit_scene_vs_reference_and_oracle(monkeypatch):
    """lit12_n12, captured from the reference's generic module: simplices (batched and loose), a Solid cube and sphere,
    point + global light, shadows, reflection depth 2 -- n = 12 is beyond the compile-time-N kernels.  Solids are there,
    so the default is composite_kernel_var_t<true> (the reference's o_hit.normal handling): the default-mode oracle to 1e-5;
    NTRACER_CLEAN_NORMALS=1 is composite_kernel_var and the clean-mode oracle.  Against the reference's own colours a
    handful of the 10 600 samples differ in either mode, as the oracle's do (test_oracle_golden.py: 5 and 9)."""
    g = fx.load("lit12_n12")
    flat = fx.flat_of(g)
    p = fx.params_of(g)
    sc = tracern.CompositeScene.from_flat(12, flat)
    sc.set_params_flat(p)
    for clean in (False, True):
        if clean:
            monkeypatch.setenv("NTRACER_CLEAN_NORMALS", "1")
        bad = total = 0
        for k, f in enumerate(g["frames"]):
            sc._set_camera_arrays(g["origins"][f], g["axes"][f])
            c = sc.colors_at(g["xs"], g["ys"], 160, 100)
            o = ob.OracleScene(12, g["origins"][f], g["axes"][f], flat=flat, params=p, clean_normals=clean).colors_at(g["xs"], g["ys"], 160, 100)
            assert np.abs(c - o).max() < TOL_ORACLE, (int(f), clean)
            d = np.abs(c - g["colors"][k]).max(axis=1)
            bad += int((d > TOL_REF).sum())
            total += len(d)
        assert bad <= 0.002 * total, (bad, clean)
        # image path
        sc._set_camera_arrays(g["origins"][7], g["axes"][7])
        img = render_host(sc, fmt_of(160, 100, fx.RGBF32))
        ref = ob.OracleScene(12, g["origins"][7], g["axes"][7], flat=flat, params=p, clean_normals=clean).render(160, 100, fx.RGBF32, threads=7)
        assert np.abs(img.view(">f4") - ref.view(">f4")).max() < TOL_ORACLE, clean


@pytest.mark.parametrize("clean", [False, True])
def test_run_time_n_kernel_has_the_feature_set_of_the_fixed_ones(monkeypatch, clean):
    """NTRACER_FORCE_VAR=1 sends the 3-D feature scene (all materials opaque: loose triangles, a Solid cube, two spheres,
    lights, shadows, reflection to depth 4 / 1 / 0) through the run-time-n kernels: composite_kernel_var_t<true> (the
    reference's o_hit.normal handling, as Solids are present) against the default-mode oracle, and with
    NTRACER_CLEAN_NORMALS=1 composite_kernel_var against the clean-mode one."""
    monkeypatch.setenv("NTRACER_FORCE_VAR", "1")
    if clean:
        monkeypatch.setenv("NTRACER_CLEAN_NORMALS", "1")
    g = fx.load("feature3d")
    flat = fx.flat_of(g, opaque=True)
    w, h = int(g["width"]), int(g["height"])
    ys, xs = np.mgrid[0:h, 0:w]
    sc = tracern.CompositeScene.from_flat(3, flat)
    sc._set_camera_arrays(g["origin"], g["axes"])
    for v in g["variants"]:
        p = fx.params_of(g, "%s__" % v)
        sc.set_params_flat(p)
        c = sc.colors_at(xs.ravel(), ys.ravel(), w, h)
        o = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p, clean_normals=clean).colors_at(xs.ravel(), ys.ravel(), w, h)
        assert np.abs(c - o).max() < TOL_ORACLE, str(v)


def test_run_time_n_transparency_kernel_vs_oracle_and_the_fixed_kernel(monkeypatch):
    """composite_kernel_var_t (transparent-hit lists, trims, compositing, shadow filtering, the exact `checked` list, frames
    in global scratch) on the scenes the compile-time-N transparency kernel is tested with: feature3d with its transparent
    materials, every variant, in both normal modes; feature5_n5 against the reference's colours; an image render and a
    multi-frame launch (blocks striding over the tiles).  The two kernels follow the same operation order: equal colours."""
    g = fx.load("feature3d")
    flat = fx.flat_of(g)
    w, h = int(g["width"]), int(g["height"])
    ys, xs = np.mgrid[0:h, 0:w]
    fixed = {}
    sc = tracern.CompositeScene.from_flat(3, flat)
    sc._set_camera_arrays(g["origin"], g["axes"])
    for v in g["variants"]:
        sc.set_params_flat(fx.params_of(g, "%s__" % v))
        fixed[str(v)] = sc.colors_at(xs.ravel(), ys.ravel(), w, h)
    g5 = fx.load("feature5_n5")
    flat5, p5 = fx.flat_of(g5), fx.params_of(g5)
    sc5 = tracern.CompositeScene.from_flat(5, flat5)
    sc5.set_params_flat(p5)
    sc5._set_camera_arrays(g5["origins"][9], g5["axes"][9])
    fixed5 = sc5.colors_at(g5["xs"], g5["ys"], 160, 100)

    monkeypatch.setenv("NTRACER_FORCE_VAR", "1")
    for clean in (False, True):
        if clean:
            monkeypatch.setenv("NTRACER_CLEAN_NORMALS", "1")
        for v in g["variants"]:
            p = fx.params_of(g, "%s__" % v)
            sc.set_params_flat(p)
            c = sc.colors_at(xs.ravel(), ys.ravel(), w, h)
            o = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p, clean_normals=clean).colors_at(xs.ravel(), ys.ravel(), w, h)
            assert np.abs(c - o).max() < TOL_ORACLE, (str(v), clean)
            if not clean:
                assert np.abs(c - fixed[str(v)]).max() <= 1e-6, str(v)
    monkeypatch.delenv("NTRACER_CLEAN_NORMALS")
    c5 = sc5.colors_at(g5["xs"], g5["ys"], 160, 100)
    assert np.abs(c5 - fixed5).max() <= 1e-6
    assert np.abs(c5 - g5["colors"][1]).max() < TOL_REF
    # image renders: one frame through nt_render, three through one device launch
    img = render_host(sc, fmt_of(w, h, fx.RGB16), strict_reference=True)
    ref = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p).render(w, h, fx.RGB16, threads=3)
    assert np.abs(img.astype(int) - ref.astype(int)).max() <= 1
    import torch
    fmt = fmt_of(w, h, fx.RGBF32)
    fb = torch.zeros((3, fmt.pitch * h), dtype=torch.uint8, device="cuda")
    o3 = np.ascontiguousarray(np.stack([g["origin"]] * 3), np.float32)
    a3 = np.ascontiguousarray(np.stack([g["axes"]] * 3), np.float32)
    o3[1, 0] += 0.25
    o3[2, 1] -= 0.3
    fst = fmt._as_struct()
    _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), fmt.pitch * h, 3, o3.ctypes.data_as(_lib.f32p),
                                                  a3.ctypes.data_as(_lib.f32p), C.byref(fst), None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = fb.cpu().numpy()
    for k in range(3):
        ref = ob.OracleScene(3, o3[k], a3[k], flat=flat, params=p).render(w, h, fx.RGBF32, threads=3)
        assert np.abs(got[k].view(">f4") - ref.reshape(-1).view(">f4")).max() < TOL_ORACLE, k


def test_eleven_dimensional_feature_scene_vs_reference(monkeypatch):
    """feature11_n11, captured from the reference's generic run-time-n module (var_geometry): transparent and reflective
    simplices, Solids (one transparent), point + global light, shadows, reflection depth 3, in eleven dimensions -- above the
    compile-time-N kernels, so composite_kernel_var_t<true>.  The reference's colours on every sample (1e-4), the
    default-mode oracle's to 1e-5; NTRACER_CLEAN_NORMALS=1: the clean-mode oracle's."""
    g = fx.load("feature11_n11")
    flat = fx.flat_of(g)
    p = fx.params_of(g)
    assert (np.asarray(flat["materials"])[:, 6] < 1).sum() == 2 and len(flat["solid_recs"]) == 2
    sc = tracern.CompositeScene.from_flat(11, flat)
    sc.set_params_flat(p)
    for clean in (False, True):
        if clean:
            monkeypatch.setenv("NTRACER_CLEAN_NORMALS", "1")
        for k, f in enumerate(g["frames"]):
            sc._set_camera_arrays(g["origins"][f], g["axes"][f])
            c = sc.colors_at(g["xs"], g["ys"], 160, 100)
            o = ob.OracleScene(11, g["origins"][f], g["axes"][f], flat=flat, params=p, clean_normals=clean).colors_at(g["xs"], g["ys"], 160, 100)
            assert np.abs(c - o).max() < TOL_ORACLE, (int(f), clean)
            if not clean:
                assert np.abs(c - g["colors"][k]).max() < TOL_REF, int(f)
    monkeypatch.delenv("NTRACER_CLEAN_NORMALS")
    sc._set_camera_arrays(g["origins"][9], g["axes"][9])
    img = render_host(sc, fmt_of(160, 100, fx.RGBF32))
    ref = ob.OracleScene(11, g["origins"][9], g["axes"][9], flat=flat, params=p).render(160, 100, fx.RGBF32, threads=7)
    assert np.abs(img.view(">f4") - ref.view(">f4")).max() < TOL_ORACLE


def test_sixteen_dimensional_feature_scene_vs_reference():
    """feature16_n16 (the reference's generic module in sixteen dimensions: transparent and reflective simplices, Solids,
    lights, shadows) through composite_kernel_var_t: the oracle's colours to 1e-5, the reference's to 1e-4."""
    g = fx.load("feature16_n16")
    flat = fx.flat_of(g)
    p = fx.params_of(g)
    sc = tracern.CompositeScene.from_flat(16, flat)
    sc.set_params_flat(p)
    for k, f in enumerate(g["frames"]):
        sc._set_camera_arrays(g["origins"][f], g["axes"][f])
        c = sc.colors_at(g["xs"], g["ys"], 160, 100)
        o = ob.OracleScene(16, g["origins"][f], g["axes"][f], flat=flat, params=p).colors_at(g["xs"], g["ys"], 160, 100)
        assert np.abs(c - o).max() < TOL_ORACLE, int(f)
        assert np.abs(c - g["colors"][k]).max() < TOL_REF, int(f)
    f = int(g["frames"][1])
    img = render_host(sc, fmt_of(160, 100, fx.RGBF32))
    ref = ob.OracleScene(16, g["origins"][f], g["axes"][f], flat=flat, params=p).render(160, 100, fx.RGBF32, threads=7)
    assert np.abs(img.view(">f4") - ref.view(">f4")).max() < TOL_ORACLE


def test_reflection_among_transparent_things_to_any_depth():
    """Two mirrors facing each other with a half-transparent, slightly reflective pane and an opaque triangle between them,
    max_reflect_depth = 12: beyond the six ray_color frames the compile-time-N transparency kernel keeps, so the launch
    goes to the run-time-n kernel, whose frame stack the host sizes (the reference recurses to any depth,
    tracer.hpp:1842-1851).  Against the oracle's plain recursion; the same scene at depth 5 (compile-time-N kernel) differs."""
    from ntracer_amd import NTracer
    nt = NTracer(3)
    mirror = ntracer_amd.Material((0.95, 0.9, 0.85), 1, 0.9, 0.4, 20)
    pane = ntracer_amd.Material((0.2, 0.5, 1.0), 0.5, 0.3, 0.6, 12)
    red = ntracer_amd.Material((1, 0.3, 0.2), 1, 0.0)
    V = nt.Vector
    protos = []
    for z, flip in ((1.5, False), (-1.5, True)):
        quad = [(-300, -300, z), (300, -300, z), (300, 300, z), (-300, 300, z)]
        if flip:
            quad = quad[::-1]
        protos.append(nt.TrianglePrototype([V(*quad[0]), V(*quad[1]), V(*quad[2])], mirror))
        protos.append(nt.TrianglePrototype([V(*quad[0]), V(*quad[2]), V(*quad[3])], mirror))
    protos.append(nt.TrianglePrototype([V(-0.4, -0.3, 0.9), V(0.5, -0.3, 0.6), V(0.1, 0.5, 0.8)], red))
    protos.append(nt.TrianglePrototype([V(-2.0, -1.5, 0.45), V(2.5, -1.2, 0.35), V(0.2, 2.4, 0.5)], pane))
    sc0 = nt.build_composite_scene(protos)
    flat = sc0._flat_description()
    flat["batch_size"] = 4
    origin = np.array([0.3, 0.2, 0.0], np.float32)
    axes = np.eye(3, dtype=np.float32)
    axes[0] = (0.995, 0.0, 0.0998)
    axes[2] = (-0.0998, 0.0, 0.995)
    params = dict(fov=0.9, shadows=1, camera_light=1, max_reflect_depth=12, bg_gradient_axis=1, ambient=[.05, .05, .05], bg1=[1, 1, 1],
                  bg2=[0, 0, 0], bg3=[0, 1, 1], point_light_pos=np.zeros((0, 3)), point_light_color=np.zeros((0, 3)),
                  global_light_dir=[[0.1, -1.0, 0.2]], global_light_color=[[0.3, 0.3, 0.3]])
    w, h = 96, 64
    ys, xs = np.mgrid[0:h, 0:w]
    o, cnt = ob.OracleScene(3, origin, axes, flat=flat, params=params).colors_at(xs.ravel(), ys.ravel(), w, h, counters=True)
    assert cnt["rays"] > 8 * w * h                        # (rays that do not end on the opaque triangle go all 12 levels)
    sc = tracern.CompositeScene.from_flat(3, flat)
    sc.set_params_flat(params)
    sc._set_camera_arrays(origin, axes)
    c = sc.colors_at(xs.ravel(), ys.ravel(), w, h)
    assert np.abs(c - o).max() < TOL_ORACLE
    img = render_host(sc, fmt_of(w, h, fx.RGBF32)).view(">f4").reshape(h, w, 3)
    assert np.abs(img - np.clip(o, 0, 1).reshape(h, w, 3)).max() < TOL_ORACLE
    params["max_reflect_depth"] = 5
    sc.set_params_flat(params)
    c5 = sc.colors_at(xs.ravel(), ys.ravel(), w, h)
    o5 = ob.OracleScene(3, origin, axes, flat=flat, params=params).colors_at(xs.ravel(), ys.ravel(), w, h)
    assert np.abs(c5 - o5).max() < TOL_ORACLE
    assert (np.abs(c5 - c).max(axis=1) > 1e-3).sum() > 0.2 * w * h


@pytest.mark.parametrize("name", ["cell600_n4", "cell120_n4", "orthoplex5_n5", "simplex7_n7", "simplex9_n9", "simplex10_n10"])
def test_polytope_vs_oracle_and_reference(name):
    g = fx.load(name)
    n = int(g["dimension"])
    w, h = int(g["width"]), int(g["height"])
    flat = fx.flat_of(g)
    sc = tracern.CompositeScene.from_flat(n, flat)
    for k, f in enumerate(g["frames"]):
        sc._set_camera_arrays(g["origins"][f], g["axes"][f])
        c = sc.colors_at(g["xs"], g["ys"], w, h)
        o = ob.OracleScene(n, g["origins"][f], g["axes"][f], flat=flat).colors_at(g["xs"], g["ys"], w, h)
        assert np.abs(c - o).max() < TOL_ORACLE, (name, int(f))
        assert np.abs(c - g["colors"][k]).max() < TOL_REF, (name, int(f))
    f0 = g["frames"][0]
    sc._set_camera_arrays(g["origins"][f0], g["axes"][f0])
    img = render_host(sc, fmt_of(160, 90, fx.RGBX8))
    assert np.abs(img.astype(int) - g["image160x90_rgbx8"].astype(int)).max() <= 1    # powf last-bit rounding


def test_cell600_full_1080p_frame_vs_oracle():
    g = fx.load("cell600_n4")
    flat = fx.flat_of(g)
    sc = tracern.CompositeScene.from_flat(4, flat)
    sc._set_camera_arrays(g["origins"][33], g["axes"][33])
    img = render_host(sc, fmt_of(1920, 1080, fx.RGBX8))
    ref = ob.OracleScene(4, g["origins"][33], g["axes"][33], flat=flat).render(1920, 1080, fx.RGBX8, threads=7)
    d = np.abs(img.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).sum() < 1e-4 * d.size


def test_cell120_full_size_properties_and_counters():
    """config 4 at 1920x1080: too slow for the oracle in full, so check (a) oracle on a sampled lattice
    (above), (b) band-decomposition invariance, (c) multi-frame launch == single launches,
    (d) the device work counters against the oracle's on the same lattice."""
    import torch
    g = fx.load("cell120_n4")
    flat = fx.flat_of(g)
    sc = tracern.CompositeScene.from_flat(4, flat)
    w, h = 1920, 1080
    fmt = fmt_of(w, h, fx.RGBX8)
    sc._set_camera_arrays(g["origins"][11], g["axes"][11])
    img = render_host(sc, fmt, collect_stats=True)
    st = sc.last_stats()
    assert st["rays"] == w * h
    re = np.empty_like(img)
    for r in range(3):
        rows = ntd.owned_rows(h, r, 3)
        buf = bytearray(len(rows) * fmt.pitch)
        assert ntracer_amd.BlockingRenderer().render(buf, fmt, sc, band_rank=r, band_world=3, compact=True)
        re[rows] = np.frombuffer(bytes(buf), np.uint8).reshape(len(rows), fmt.pitch)
    assert np.array_equal(re, img)
    # multi-frame launch
    frames = [11, 52]
    fb = torch.zeros((2, h * fmt.pitch), dtype=torch.uint8, device="cuda")
    o = np.ascontiguousarray(g["origins"][frames], np.float32)
    a = np.ascontiguousarray(g["axes"][frames], np.float32)
    st_ = fmt._as_struct()
    _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), h * fmt.pitch, 2,
                                                  o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(st_),
                                                  None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(fb[0].cpu().numpy().reshape(h, fmt.pitch), img)
    sc._set_camera_arrays(g["origins"][52], g["axes"][52])
    assert np.array_equal(fb[1].cpu().numpy().reshape(h, fmt.pitch), render_host(sc, fmt))
    # counters: same tree walk as the oracle (branches/leaves identical; the device's mailbox is small, so it
    # tests at least as many simplices) -- for the reference's exact walk (strict) and for the default walk that
    # drops cells beyond the current hit
    sc._set_camera_arrays(g["origins"][0], g["axes"][0])
    lat = fmt_of(w, h, fx.RGBX8)
    per = lambda d, k: d[k] / d["rays"]
    walks = {}
    for strict in (True, False):
        img0 = render_host(sc, lat, collect_stats=True, strict_reference=strict)
        st = sc.last_stats()
        _, oc = ob.OracleScene(4, g["origins"][0], g["axes"][0], flat=flat, prune=not strict).colors_at(g["xs"], g["ys"], w, h, counters=True)
        assert abs(per(st, "branches") - per(oc, "branches")) < 0.05 * per(oc, "branches")
        assert abs(per(st, "leaves") - per(oc, "leaves")) < 0.05 * per(oc, "leaves")
        assert abs(per(st, "hits") - per(oc, "hits")) < 0.02
        assert per(st, "simplex_tests") >= 0.95 * per(oc, "simplex_tests")
        walks[strict] = (img0, per(st, "simplex_tests"))
    assert np.array_equal(walks[True][0], walks[False][0])
    assert walks[False][1] < walks[True][1]


def test_pruned_walk_renders_the_reference_walks_bytes():
    """The default closest-hit walk skips k-d cells that begin beyond the current hit (nt_beyond_hit); with
    strict_reference the kernels visit exactly the cells the reference visits.  Same bytes, on every kernel:
    120-cell frames where the reference's walk runs on to the far end of the scene for the central rays (packet
    kernel), the lit / reflective variant (packet + per-lane secondary rays), and the mixed feature scene (tile
    kernel with solids and loose triangles)."""
    g = fx.load("cell120_n4")
    sc = tracern.CompositeScene.from_flat(4, fx.flat_of(g))
    fmt = fmt_of(1920, 1080, fx.RGBX8)
    for f in (40, 120):
        sc._set_camera_arrays(g["origins"][f], g["axes"][f])
        assert np.array_equal(render_host(sc, fmt, strict_reference=True), render_host(sc, fmt, strict_reference=False))
    g = fx.load("cell600_n4")
    flat = fx.flat_of(g)
    m = np.array(flat["materials"], np.float32).copy()
    m[:, 7] = 0.3
    flat["materials"] = m
    sc = tracern.CompositeScene.from_flat(4, flat)
    sc.set_params_flat(dict(fov=0.8, shadows=1, camera_light=1, max_reflect_depth=2, bg_gradient_axis=1,
                            ambient=[.02, .02, .03], bg1=[1, 1, 1], bg2=[0, 0, 0], bg3=[0, 1, 1],
                            point_light_pos=[[20.0, 15.0, -25.0, 5.0]], point_light_color=[[900.0, 800.0, 700.0]],
                            global_light_dir=[[0.2, -0.9, 0.3, 0.1]], global_light_color=[[.4, .4, .5]]))
    sc._set_camera_arrays(g["origins"][5], g["axes"][5])
    fmt = fmt_of(480, 270, fx.RGBF32)
    assert np.array_equal(render_host(sc, fmt, strict_reference=True), render_host(sc, fmt, strict_reference=False))
    g = fx.load("feature3d")
    flat = fx.flat_of(g, opaque=True)                  # transparency always walks strictly
    for v in g["variants"]:
        sc = tracern.CompositeScene.from_flat(3, flat)
        sc.set_params_flat(fx.params_of(g, "%s__" % v))
        sc._set_camera_arrays(g["origin"], g["axes"])
        fmt = fmt_of(320, 240, fx.RGBF32)
        assert np.array_equal(render_host(sc, fmt, strict_reference=True), render_host(sc, fmt, strict_reference=False))


def test_three_composite_kernels_render_identical_frames(monkeypatch):
    """Lean scenes can be rendered by the wave-uniform packet kernel (default), the persistent per-lane kernel
    with ballot-driven ray refill, or the plain per-lane tile kernel.  All walk the reference tree in the
    reference order, so their frames must be byte-identical (and equal the oracle's up to powf rounding)."""
    g = fx.load("cell600_n4")
    flat = fx.flat_of(g)
    fmt = fmt_of(333, 217, fx.RGBF32)          # ragged size, float channels: colours compared bit for bit
    frames = {}
    for choice in ("0", "1", "2"):
        monkeypatch.setenv("NTRACER_COMPOSITE_KERNEL", choice)
        sc = tracern.CompositeScene.from_flat(4, flat)
        sc._set_camera_arrays(g["origins"][77], g["axes"][77])
        frames[choice] = render_host(sc, fmt)
    assert np.array_equal(frames["0"], frames["1"])
    assert np.array_equal(frames["0"], frames["2"])
    ref = ob.OracleScene(4, g["origins"][77], g["axes"][77], flat=flat).render(333, 217, fx.RGBF32, threads=7)
    got = frames["0"].view(">f4")
    assert np.abs(got - ref.view(">f4")).max() < TOL_ORACLE


def test_lit_reflective_polytope_packet_vs_tile_vs_oracle(monkeypatch):
    """Lights, shadows and reflection on a batches-only scene: the packet kernel finds the primary hits and shades
    per lane (shadow / reflection rays walk the tree per lane); the per-lane tile kernel does everything per lane.
    Both must produce the same bytes, and the oracle's colours."""
    g = fx.load("cell600_n4")
    flat = fx.flat_of(g)
    m = np.array(flat["materials"], np.float32).copy()
    m[:, 7] = 0.3                                      # reflective
    flat["materials"] = m
    params = dict(fov=0.8, shadows=1, camera_light=1, max_reflect_depth=2, bg_gradient_axis=1,
                  ambient=[.02, .02, .03], bg1=[1, 1, 1], bg2=[0, 0, 0], bg3=[0, 1, 1],
                  point_light_pos=[[20.0, 15.0, -25.0, 5.0]], point_light_color=[[900.0, 800.0, 700.0]],
                  global_light_dir=[[0.2, -0.9, 0.3, 0.1]], global_light_color=[[.4, .4, .5]])
    fmt = fmt_of(240, 135, fx.RGBF32)
    frames = {}
    for choice in ("0", "2"):
        monkeypatch.setenv("NTRACER_COMPOSITE_KERNEL", choice)
        sc = tracern.CompositeScene.from_flat(4, flat)
        sc.set_params_flat(params)
        sc._set_camera_arrays(g["origins"][5], g["axes"][5])
        frames[choice] = render_host(sc, fmt)
    assert np.array_equal(frames["0"], frames["2"])
    ref = ob.OracleScene(4, g["origins"][5], g["axes"][5], flat=flat, params=params).render(240, 135, fx.RGBF32, threads=7)
    assert np.abs(frames["0"].view(">f4") - ref.view(">f4")).max() < TOL_ORACLE
    assert frames["0"].view(">f4").max() > 0.5


def test_multi_frame_launches_in_chunks(monkeypatch):
    """Multi-frame launches of composite scenes go through scratch buffers sized in frames (plane numerators; for
    lit scenes also the primary hits of the two-pass render) and are cut into chunks when those are too small.
    Forcing one- and two-frame chunks must not change a byte, lit or not."""
    import torch
    g = fx.load("cell600_n4")
    fmt = fmt_of(320, 180, fx.RGBF32)
    frames = [3, 40, 77, 111, 150]
    o = np.ascontiguousarray(g["origins"][frames], np.float32)
    a = np.ascontiguousarray(g["axes"][frames], np.float32)
    st_ = fmt._as_struct()
    for lit in (False, True):
        flat = fx.flat_of(g)
        sc = tracern.CompositeScene.from_flat(4, flat)
        if lit:
            sc.set_params_flat(dict(fov=0.8, shadows=1, camera_light=1, max_reflect_depth=2, bg_gradient_axis=1,
                                    ambient=[.02, .02, .03], bg1=[1, 1, 1], bg2=[0, 0, 0], bg3=[0, 1, 1],
                                    point_light_pos=[[20.0, 15.0, -25.0, 5.0]], point_light_color=[[900.0, 800.0, 700.0]],
                                    global_light_dir=[[0.2, -0.9, 0.3, 0.1]], global_light_color=[[.4, .4, .5]]))
        out = {}
        for chunk in ("", "1", "2"):
            if chunk:
                monkeypatch.setenv("NTRACER_CHUNK_FRAMES", chunk)
            else:
                monkeypatch.delenv("NTRACER_CHUNK_FRAMES", raising=False)
            fb = torch.zeros((len(frames), fmt.height * fmt.pitch), dtype=torch.uint8, device="cuda")
            _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), fmt.height * fmt.pitch, len(frames),
                                                          o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(st_),
                                                          None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            out[chunk] = fb.cpu().numpy()
        assert np.array_equal(out[""], out["1"]) and np.array_equal(out[""], out["2"])
        monkeypatch.delenv("NTRACER_CHUNK_FRAMES", raising=False)
        sc._set_camera_arrays(g["origins"][77], g["axes"][77])
        assert np.array_equal(out[""][2].reshape(fmt.height, fmt.pitch), render_host(sc, fmt))
        assert out[""].view(">f4").max() > 0.3


def test_rebuilt_tree_renders_the_same_bytes():
    """CompositeScene.with_rebuilt_tree keeps the primitive tables and replaces the k-d tree by one from the native
    builder: nearest hits do not depend on the tree, so full frames come out the same -- on the reference-built
    600-cell and 120-cell, default and strict walks -- except where a ray meets the shared edge of two simplices at
    exactly the same distance: the first one tested wins (tracer.hpp:585, strict <), and the order is the tree's."""
    for name, frames in (("cell600_n4", (5, 77)), ("cell120_n4", (40,))):
        g = fx.load(name)
        sc = tracern.CompositeScene.from_flat(4, fx.flat_of(g))
        reb = sc.with_rebuilt_tree()
        assert len(reb._flat["batch_recs"]) == len(sc._flat["batch_recs"])
        fmt = fmt_of(1920, 1080, fx.RGBX8)
        for f in frames:
            sc._set_camera_arrays(g["origins"][f], g["axes"][f])
            reb._set_camera_arrays(g["origins"][f], g["axes"][f])
            a = render_host(sc, fmt).reshape(1080, 1920, 4)
            b = render_host(reb, fmt).reshape(1080, 1920, 4)
            ties = (a != b).any(axis=2).sum()
            assert ties <= 2e-4 * 1920 * 1080, (name, f, int(ties))
            assert np.array_equal(b, render_host(reb, fmt, strict_reference=True).reshape(1080, 1920, 4)), (name, f)


def test_scenes_with_solids_and_loose_triangles_take_the_packet_walk(monkeypatch):
    """Image renders of opaque scenes whose leaves hold unbatched triangles and solids go through the packet
    kernel for the primary hits (per-lane tests on uniformly addressed records) and the general kernel for the
    shading.  Same bytes as the all-per-lane tile kernel, and the oracle's colours, on every variant of the feature
    scene (lights, shadows, reflection) and on the 10-D simplex (2 batches + 3 loose triangles)."""
    g = fx.load("feature3d")
    flat = fx.flat_of(g, opaque=True)
    w, h = int(g["width"]), int(g["height"])
    fmt = fmt_of(w, h, fx.RGBF32)
    # (by default a scene with Solids is rendered with the reference's o_hit.normal handling by composite_kernel_t<N, true>;
    # the packet walk serves the intended-semantics mode)
    monkeypatch.setenv("NTRACER_CLEAN_NORMALS", "1")
    for v in g["variants"]:
        p = fx.params_of(g, "%s__" % v)
        frames = {}
        for choice in ("0", "2"):
            monkeypatch.setenv("NTRACER_COMPOSITE_KERNEL", choice)
            sc = tracern.CompositeScene.from_flat(3, flat)
            sc.set_params_flat(p)
            sc._set_camera_arrays(g["origin"], g["axes"])
            frames[choice] = render_host(sc, fmt)
        assert np.array_equal(frames["0"], frames["2"]), str(v)
        ref = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p, clean_normals=True).render(w, h, fx.RGBF32, threads=7)
        assert np.abs(frames["0"].view(">f4") - ref.view(">f4")).max() < TOL_ORACLE, str(v)
    g = fx.load("simplex10_n10")
    flat = fx.flat_of(g)
    assert len(flat["tri_recs"]) == 3
    fmt = fmt_of(200, 120, fx.RGBF32)
    frames = {}
    for choice in ("0", "2"):
        monkeypatch.setenv("NTRACER_COMPOSITE_KERNEL", choice)
        sc = tracern.CompositeScene.from_flat(10, flat)
        sc._set_camera_arrays(g["origins"][9], g["axes"][9])
        frames[choice] = render_host(sc, fmt)
    assert np.array_equal(frames["0"], frames["2"])


@pytest.mark.parametrize("band_rows,world", [(8, 8), (16, 4), (8, 3)])
def test_band_heights_other_than_32(band_rows, world):
    """nt_render_opts.band_rows: the ranks' compact buffers, de-interleaved with the same band height, give the
    whole frame -- BoxScene (64x4-pixel blocks) and a composite scene (16x16-pixel quads of the packet kernel)."""
    g = fx.load("cell600_n4")
    scenes = [tracern.BoxScene(6), tracern.CompositeScene.from_flat(4, fx.flat_of(g)), tracern.BoxScene(12)]
    gb = fx.load("box_n6_1920x1080")
    scenes[0]._set_camera_arrays(gb["origins"][21], gb["axes"][21])
    scenes[1]._set_camera_arrays(g["origins"][33], g["axes"][33])
    g12 = fx.load("box_n12_320x200")                     # (run-time-n rows kernel)
    scenes[2]._set_camera_arrays(g12["origins"][g12["frames"][1]], g12["axes"][g12["frames"][1]])
    fmt = fmt_of(500, 301, fx.RGBX8)
    for sc in scenes:
        whole = render_host(sc, fmt)
        re = np.zeros_like(whole)
        for r in range(world):
            rows = ntd.owned_rows(fmt.height, r, world, band_rows)
            buf = bytearray(len(rows) * fmt.pitch)
            assert ntracer_amd.BlockingRenderer().render(buf, fmt, sc, band_rank=r, band_world=world, compact=True, band_rows=band_rows)
            re[rows] = np.frombuffer(bytes(buf), np.uint8).reshape(len(rows), fmt.pitch)
        assert np.array_equal(re, whole)


@pytest.mark.parametrize("w,h,pitch", [(64, 8, 0), (333, 9, 1000), (333, 9, 0), (1920, 16, 0), (5, 3, 16)])
@pytest.mark.parametrize("rev", [False, True])
def test_three_byte_pixels_are_packed_across_lanes(w, h, pitch, rev):
    """RGB24: four neighbouring lanes turn their 3-byte pixels into three dword stores when the rows are 4-byte
    aligned (store_word32_narrow), byte stores otherwise (odd pitches, groups cut by the image edge).  Same bytes as
    the oracle either way: BoxScene (exact) and a composite scene through the packet and tile kernels (+-1)."""
    rgb24 = [(8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1)]
    gb = fx.load("box_n6_1920x1080")
    fmt = fmt_of(w, h, rgb24, pitch, rev)
    sc = tracern.BoxScene(6)
    sc._set_camera_arrays(gb["origins"][17], gb["axes"][17])
    buf = bytearray(b"\xa5" * (fmt.pitch * h))
    assert ntracer_amd.BlockingRenderer().render(buf, fmt, sc)
    got = np.frombuffer(bytes(buf), np.uint8).reshape(h, fmt.pitch)
    ref = ob.OracleScene(6, gb["origins"][17], gb["axes"][17]).render(w, h, rgb24, pitch=fmt.pitch, reversed_=rev)
    assert np.array_equal(got[:, :w * 3], ref[:, :w * 3])
    assert (got[:, w * 3:] == 0xa5).all()                  # padding bytes of the rows are left alone
    g = fx.load("cell600_n4")
    flat = fx.flat_of(g)
    cs = tracern.CompositeScene.from_flat(4, flat)
    cs._set_camera_arrays(g["origins"][33], g["axes"][33])
    buf = bytearray(fmt.pitch * h)
    assert ntracer_amd.BlockingRenderer().render(buf, fmt, cs)
    got = np.frombuffer(bytes(buf), np.uint8).reshape(h, fmt.pitch)[:, :w * 3]
    ref = ob.OracleScene(4, g["origins"][33], g["axes"][33], flat=flat).render(w, h, rgb24, pitch=fmt.pitch, reversed_=rev)[:, :w * 3]
    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1


@pytest.mark.parametrize("w,h,pitch,rev", [(64, 8, 0, False), (333, 5, 0, True), (333, 5, 2002, False), (7, 3, 48, True)])
def test_six_byte_pixels_are_packed_across_lane_pairs(w, h, pitch, rev):
    """RGB48 (three 16-bit channels, the reference's video-export format): lane pairs write three dwords per two
    pixels on aligned rows (store_word64_narrow); the bytes are the oracle's."""
    rgb48 = [(16, 1, 0, 0), (16, 0, 1, 0), (16, 0, 0, 1)]
    gb = fx.load("box_n6_1920x1080")
    fmt = fmt_of(w, h, rgb48, pitch, rev)
    sc = tracern.BoxScene(6)
    sc._set_camera_arrays(gb["origins"][17], gb["axes"][17])
    buf = bytearray(b"\x5a" * (fmt.pitch * h))
    assert ntracer_amd.BlockingRenderer().render(buf, fmt, sc)
    got = np.frombuffer(bytes(buf), np.uint8).reshape(h, fmt.pitch)
    ref = ob.OracleScene(6, gb["origins"][17], gb["axes"][17]).render(w, h, rgb48, pitch=fmt.pitch, reversed_=rev)
    assert np.array_equal(got[:, :w * 6], ref[:, :w * 6])
    assert (got[:, w * 6:] == 0x5a).all()
    g = fx.load("cell600_n4")
    flat = fx.flat_of(g)
    cs = tracern.CompositeScene.from_flat(4, flat)
    cs._set_camera_arrays(g["origins"][33], g["axes"][33])
    buf = bytearray(fmt.pitch * h)
    assert ntracer_amd.BlockingRenderer().render(buf, fmt, cs)
    got = np.frombuffer(bytes(buf), np.uint8).reshape(h, fmt.pitch)[:, :w * 6].reshape(h, w, 3, 2)
    ref = ob.OracleScene(4, g["origins"][33], g["axes"][33], flat=flat).render(w, h, rgb48, pitch=fmt.pitch, reversed_=rev)[:, :w * 6].reshape(h, w, 3, 2)
    to16 = (lambda a: a[..., 1].astype(int) * 256 + a[..., 0]) if rev else (lambda a: a[..., 0].astype(int) * 256 + a[..., 1])
    gv, rv = to16(got), to16(ref)
    if rev:                                              # reversed pixels: channel order flips too
        gv, rv = gv[..., ::-1], rv[..., ::-1]
    assert np.abs(gv - rv).max() <= 2                    # powf rounding, in 16-bit units


def test_bench_workload_every_frame_equals_the_oracle():
    """The bench's step itself -- BoxScene(6), 1920x1080 RGBX8, the 160 cameras of the rotation in ONE launch -- against
    the oracle, frame by frame, byte for byte (the oracle uses the host's cores: a few seconds on the GPU box)."""
    import os
    import torch
    g = fx.load("box_n6_1920x1080")
    w, h = 1920, 1080
    fmt = fmt_of(w, h, fx.RGBX8)
    sc = tracern.BoxScene(6)
    o = np.ascontiguousarray(g["origins"], np.float32)
    a = np.ascontiguousarray(g["axes"], np.float32)
    nf = len(o)
    fb = torch.zeros((nf, h * fmt.pitch), dtype=torch.uint8, device="cuda")
    st_ = fmt._as_struct()
    _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), h * fmt.pitch, nf,
                                                  o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(st_),
                                                  None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = fb.cpu().numpy().reshape(nf, h, fmt.pitch)
    osc = ob.OracleScene(6, o[0], a[0])
    threads = max(1, min(63, (os.cpu_count() or 2) - 1))
    step = 1 if threads >= 16 else 8                     # a small host checks every 8th frame
    for f in range(0, nf, step):
        osc.set_camera(o[f], a[f])
        ref = osc.render(w, h, fx.RGBX8, threads=threads)
        assert np.array_equal(got[f], ref), f


@pytest.mark.parametrize("n,chans", [(4, "rgbx8"), (8, "rgbx8"), (10, "rgbf32")])
def test_tall_launches_take_sixty_four_rows_a_wave(n, chans):
    """Launches of 512 rows or more with 32k tiles to go round run box_tile_kernel<N, F32, 64, 1> (one wave per 64 x 64
    tile); the bench frames cover that for n = 3 and 6 -- here 72 random cameras at 1080p for other dimensions and the fp32
    format, every sixth frame against the oracle byte for byte."""
    import torch
    rng = np.random.default_rng(700 + n)
    w, h, nf = 1920, 1080, 72
    fmt = fmt_of(w, h, fx.RGBX8 if chans == "rgbx8" else fx.RGBF32)
    origins, axes = [], []
    for k in range(nf):
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        q = np.ascontiguousarray(q, np.float32)
        dist = float(rng.choice([1.6, 2.5, 4.0, 7.0]))
        origins.append((-q[2] * np.float32(dist) + np.float32(rng.uniform(-0.5, 0.5)) * q[0]).astype(np.float32))
        axes.append(q)
    o = np.ascontiguousarray(np.stack(origins), np.float32)
    a = np.ascontiguousarray(np.stack(axes), np.float32)
    sc = tracern.BoxScene(n)
    fb = torch.zeros((nf, h * fmt.pitch), dtype=torch.uint8, device="cuda")
    st_ = fmt._as_struct()
    _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), h * fmt.pitch, nf, o.ctypes.data_as(_lib.f32p),
                                                  a.ctypes.data_as(_lib.f32p), C.byref(st_), None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = fb.cpu().numpy().reshape(nf, h, fmt.pitch)
    osc = ob.OracleScene(n, o[0], a[0])
    for f in range(0, nf, 6):
        osc.set_camera(o[f], a[f])
        ref = osc.render(w, h, fx.RGBX8 if chans == "rgbx8" else fx.RGBF32, threads=7)
        assert np.array_equal(got[f], ref), (n, f)


def test_bench_workload_in_bands_and_in_float_channels():
    """The large-launch tile shape (sixteen rows a lane) in the two other guises the bench uses: (1) one rank's bands of an
    8-GPU run -- rows dealt in bands of 8, compact buffer -- must be the corresponding rows of the whole frames (which
    test_bench_workload_every_frame_equals_the_oracle pins to the oracle); (2) three fp32 channels: whole frames byte for
    byte against the oracle."""
    import os
    import torch
    g = fx.load("box_n6_1920x1080")
    w, h = 1920, 1080
    sc = tracern.BoxScene(6)
    o = np.ascontiguousarray(g["origins"], np.float32)
    a = np.ascontiguousarray(g["axes"], np.float32)
    nf = len(o)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch(fmt, frames, opts, rows):
        fb = torch.zeros((len(frames), rows * fmt.pitch), dtype=torch.uint8, device="cuda")
        st_ = fmt._as_struct()
        of, af = np.ascontiguousarray(o[frames]), np.ascontiguousarray(a[frames])
        _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), rows * fmt.pitch, len(frames), of.ctypes.data_as(_lib.f32p),
                                                      af.ctypes.data_as(_lib.f32p), C.byref(st_), C.byref(opts) if opts is not None else None, stream))
        torch.cuda.synchronize()
        return fb.cpu().numpy().reshape(len(frames), rows, fmt.pitch)

    fmt = fmt_of(w, h, fx.RGBX8)
    whole = launch(fmt, list(range(nf)), None, h)
    for rank in (0, 5):
        opts = _lib.NtRenderOpts()
        opts.device, opts.band_rank, opts.band_world, opts.band_rows, opts.compact = 0, rank, 8, 8, 1
        rows = ntd.owned_rows(h, rank, 8, 8)
        part = launch(fmt, list(range(nf)), opts, len(rows))
        assert np.array_equal(part, whole[:, rows, :]), rank
    frames = list(range(0, nf, 4))                        # 40 frames: still the large-launch shape
    fmt32 = fmt_of(w, h, fx.RGBF32)
    got = launch(fmt32, frames, None, h)
    threads = max(1, min(63, (os.cpu_count() or 2) - 1))
    osc = ob.OracleScene(6, o[0], a[0])
    for k in (0, 7, 23) if threads < 16 else range(0, len(frames), 3):
        osc.set_camera(o[frames[k]], a[frames[k]])
        assert np.array_equal(got[k], osc.render(w, h, fx.RGBF32, threads=threads)), frames[k]


@pytest.mark.parametrize("frame", [0, 40, 93])
def test_cell120_full_1080p_frames_vs_oracle(frame):
    """config 4 at full size against the oracle on whole frames (the oracle walks with prune_beyond_hit, which its own
    test shows to be pixel-identical to the reference's walk and is an order of magnitude faster)."""
    import os
    g = fx.load("cell120_n4")
    flat = fx.flat_of(g)
    sc = tracern.CompositeScene.from_flat(4, flat)
    sc._set_camera_arrays(g["origins"][frame], g["axes"][frame])
    img = render_host(sc, fmt_of(1920, 1080, fx.RGBX8))
    threads = max(1, min(63, (os.cpu_count() or 2) - 1))
    ref = ob.OracleScene(4, g["origins"][frame], g["axes"][frame], flat=flat, prune=True).render(1920, 1080, fx.RGBX8, threads=threads)
    d = np.abs(img.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).sum() < 1e-4 * d.size


@pytest.mark.parametrize("order", [(0, 1, 2), (2, 0, 1)])
@pytest.mark.parametrize("rev", [False, True])
def test_three_float_channels_fast_path(order, rev):
    """fp32 x 3 pixels whose channels are plain components (any order) take a short epilogue: clamp, big-endian
    float bits, three dword stores.  Bytes equal the oracle's generic packing."""
    chans = [(32,) + tuple(1.0 if c == k else 0.0 for c in range(3)) + (0.0, True) for k in order]
    gb = fx.load("box_n6_1920x1080")
    sc = tracern.BoxScene(6)
    sc._set_camera_arrays(gb["origins"][29], gb["axes"][29])
    fmt = fmt_of(401, 33, chans, 0, rev)
    got = render_host(sc, fmt)
    ref = ob.OracleScene(6, gb["origins"][29], gb["axes"][29]).render(401, 33, chans, reversed_=rev)
    assert np.array_equal(got, ref)
    assert got.max() > 0


def test_config4_whole_rotation_default_walk_equals_reference_walk():
    """All 160 cameras of the 120-cell rotation at 1920x1080: the default walk (cells beyond the hit dropped) and the
    reference's exact walk (strict_reference) must produce the same bytes, frame for frame; and so must the 600-cell."""
    import torch
    for name in ("cell120_n4", "cell600_n4"):
        g = fx.load(name)
        sc = tracern.CompositeScene.from_flat(4, fx.flat_of(g))
        fmt = fmt_of(1920, 1080, fx.RGBX8)
        st_ = fmt._as_struct()
        o = np.ascontiguousarray(g["origins"], np.float32)
        a = np.ascontiguousarray(g["axes"], np.float32)
        chunk = 40
        for f0 in range(0, len(o), chunk):
            out = []
            for strict in (0, 1):
                opts = _lib.NtRenderOpts()
                opts.device = -1
                opts.band_world = 1
                opts.strict_reference = strict
                fb = torch.zeros((chunk, 1080 * fmt.pitch), dtype=torch.uint8, device="cuda")
                _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), 1080 * fmt.pitch, chunk,
                                                              o[f0:f0 + chunk].ctypes.data_as(_lib.f32p), a[f0:f0 + chunk].ctypes.data_as(_lib.f32p),
                                                              C.byref(st_), C.byref(opts), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                torch.cuda.synchronize()
                out.append(fb)
            assert torch.equal(out[0], out[1]), (name, f0)
            assert int(out[0].max()) > 0


def test_reference_known_answer_scene_on_gpu():
    """lib/ntracer/tests/test.py:303-363 through the GPU: a camera at the test ray's origin looking along
    its direction; the centre pixel must be shaded exactly as the oracle shades the hit on primitives[4]."""
    ka = fx.known_answer()
    flat = fx.known_answer_flat(ka)
    d = np.asarray(ka["ray"]["direction"], np.float32)
    a = np.array([0, 1, 0], np.float32)
    right = np.cross(a, d)
    right /= np.linalg.norm(right)
    up = np.cross(d, right)
    axes = np.stack([right, up, d]).astype(np.float32)
    sc = tracern.CompositeScene.from_flat(3, flat)
    sc._set_camera_arrays(ka["ray"]["origin"], axes)
    osc = ob.OracleScene(3, ka["ray"]["origin"], axes, flat=flat)
    hit = osc.kd_intersects(ka["ray"]["origin"], osc.primary_dir(32, 32, 64, 64))
    assert hit is not None and hit["index"] == ka["expect"]["primitive_index"]
    ys, xs = np.mgrid[0:64, 0:64]
    c = sc.colors_at(xs.ravel(), ys.ravel(), 64, 64)
    o = osc.colors_at(xs.ravel(), ys.ravel(), 64, 64)
    assert np.abs(c - o).max() < TOL_ORACLE
    assert c.reshape(64, 64, 3)[32, 32].max() > 0        # a hit, not background


def test_lights_shadows_reflection_solids_vs_oracle(monkeypatch):
    """feature scene with every material made opaque (transparency: see test_transparency_vs_oracle): lights,
    shadows incl. the far-child quirk, reflection to depth 4/1/0, Solid cube + spheres, unbatched triangles.
    Default = the reference's behaviour, o_hit.normal aliasing included (tracer.hpp:1001,1020,133,138): against the oracle
    in its default mode, the one pinned to the reference's goldens.  NTRACER_CLEAN_NORMALS=1 = the intended semantics:
    against the oracle's clean mode."""
    g = fx.load("feature3d")
    flat = fx.flat_of(g, opaque=True)
    w, h = int(g["width"]), int(g["height"])
    ys, xs = np.mgrid[0:h, 0:w]
    sc = tracern.CompositeScene.from_flat(3, flat)
    sc._set_camera_arrays(g["origin"], g["axes"])
    for clean in (False, True):
        if clean:
            monkeypatch.setenv("NTRACER_CLEAN_NORMALS", "1")
        differ = 0
        for v in g["variants"]:
            p = fx.params_of(g, "%s__" % v)
            sc.set_params_flat(p)
            c = sc.colors_at(xs.ravel(), ys.ravel(), w, h)
            o = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p, clean_normals=clean).colors_at(xs.ravel(), ys.ravel(), w, h)
            assert np.abs(c - o).max() < TOL_ORACLE, (str(v), clean)
            other = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p, clean_normals=not clean).colors_at(xs.ravel(), ys.ravel(), w, h)
            differ += int((np.abs(c - other).max(axis=1) > TOL_REF).sum())
            # strict_reference changes nothing here (scenes with Solids are always walked strictly)
            img = render_host(sc, fmt_of(w, h, fx.RGBF32), strict_reference=True).view(">f4").reshape(h, w, 3)
            assert np.abs(img - np.clip(o, 0, 1).reshape(h, w, 3)).max() < TOL_ORACLE, (str(v), clean)      # (channels clamp)
        assert differ > 0            # the two modes are not the same thing on this scene


def test_transparency_vs_oracle(monkeypatch):
    """The full feature scene: transparent and transparent+reflective materials on top of everything above --
    transparent-hit lists with the reference's trims (tracer.hpp:1084,1228), back-to-front compositing
    (:1870-1880), shadow filtering through transparent blockers (:1755-1763), reflection off transparent
    surfaces.  Against the default-mode oracle (the reference's o_hit.normal aliasing reproduced) and directly against
    the reference's own colours, which then differ on no more pixels than the oracle's do (test_oracle_golden.py)."""
    g = fx.load("feature3d")
    flat = fx.flat_of(g)
    w, h = int(g["width"]), int(g["height"])
    ys, xs = np.mgrid[0:h, 0:w]
    sc = tracern.CompositeScene.from_flat(3, flat)
    sc._set_camera_arrays(g["origin"], g["axes"])
    for v in g["variants"]:
        p = fx.params_of(g, "%s__" % v)
        sc.set_params_flat(p)
        c = sc.colors_at(xs.ravel(), ys.ravel(), w, h)
        o = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p).colors_at(xs.ravel(), ys.ravel(), w, h)
        d = np.abs(c - o).max(axis=1)
        assert d.max() < TOL_ORACLE, (str(v), float(d.max()), int((d > TOL_ORACLE).sum()))
        dr = np.abs(c.reshape(h, w, 3) - g["%s__colors" % v]).max(axis=2)
        assert (dr > TOL_REF).sum() <= 0.005 * dr.size, (str(v), int((dr > TOL_REF).sum()))
        assert np.median(dr) < 1e-6
    img = render_host(sc, fmt_of(w, h, fx.RGB16), strict_reference=True)
    ref = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p).render(w, h, fx.RGB16, threads=3)
    assert np.abs(img.astype(int) - ref.astype(int)).max() <= 1
    # a multi-frame device launch takes the same path (blocks striding over the tiles of all frames)
    import torch
    fmt = fmt_of(w, h, fx.RGBF32)
    fb = torch.zeros((3, fmt.pitch * h), dtype=torch.uint8, device="cuda")
    o3 = np.ascontiguousarray(np.stack([g["origin"]] * 3), np.float32)
    a3 = np.ascontiguousarray(np.stack([g["axes"]] * 3), np.float32)
    o3[1, 0] += 0.25
    fst = fmt._as_struct()
    _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), fmt.pitch * h, 3, o3.ctypes.data_as(_lib.f32p),
                                                  a3.ctypes.data_as(_lib.f32p), C.byref(fst), None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = fb.cpu().numpy()
    for k in range(3):
        ref = ob.OracleScene(3, o3[k], a3[k], flat=flat, params=p).render(w, h, fx.RGBF32, threads=3)
        assert np.abs(got[k].view(">f4") - ref.reshape(-1).view(">f4")).max() < TOL_ORACLE, k
    # the intended semantics remain available
    monkeypatch.setenv("NTRACER_CLEAN_NORMALS", "1")
    for v in g["variants"]:
        p = fx.params_of(g, "%s__" % v)
        sc.set_params_flat(p)
        c = sc.colors_at(xs.ravel(), ys.ravel(), w, h)
        o = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=p, clean_normals=True).colors_at(xs.ravel(), ys.ravel(), w, h)
        assert np.abs(c - o).max() < TOL_ORACLE, str(v)


def test_reflection_deeper_than_the_level_stack(monkeypatch):
    """Two mirrors facing each other (reflectivity 0.9) and a camera between them: rays bounce until max_reflect_depth,
    here 40 -- the reference recurses to any depth (tracer.hpp:1842-1851); the kernels keep 16 levels exactly nested and
    fold deeper ones into a running affine pair.  Against the oracle's plain recursion, in three dimensions (compile-time-N
    kernels) and through the run-time-n kernel."""
    from ntracer_amd import NTracer
    nt = NTracer(3)
    mirror = ntracer_amd.Material((0.95, 0.9, 0.85), 1, 0.9, 0.4, 20)
    red = ntracer_amd.Material((1, 0.3, 0.2), 1, 0.0)
    V = nt.Vector
    protos = []
    for z, flip in ((1.5, False), (-1.5, True)):
        quad = [(-300, -300, z), (300, -300, z), (300, 300, z), (-300, 300, z)]
        if flip:
            quad = quad[::-1]
        protos.append(nt.TrianglePrototype([V(*quad[0]), V(*quad[1]), V(*quad[2])], mirror))
        protos.append(nt.TrianglePrototype([V(*quad[0]), V(*quad[2]), V(*quad[3])], mirror))
    protos.append(nt.TrianglePrototype([V(-0.4, -0.3, 0.9), V(0.5, -0.3, 0.6), V(0.1, 0.5, 0.8)], red))
    sc = nt.build_composite_scene(protos)
    flat = sc._flat_description()
    flat["batch_size"] = 4
    origin = np.array([0.3, 0.2, 0.0], np.float32)
    axes = np.eye(3, dtype=np.float32)
    axes[0] = (0.995, 0.0, 0.0998)
    axes[2] = (-0.0998, 0.0, 0.995)
    params = dict(fov=0.9, shadows=0, camera_light=1, max_reflect_depth=40, bg_gradient_axis=1, ambient=[.05, .05, .05], bg1=[1, 1, 1],
                  bg2=[0, 0, 0], bg3=[0, 1, 1], point_light_pos=np.zeros((0, 3)), point_light_color=np.zeros((0, 3)),
                  global_light_dir=[[0.1, -1.0, 0.2]], global_light_color=[[0.3, 0.3, 0.3]])
    w, h = 96, 64
    ys, xs = np.mgrid[0:h, 0:w]
    osc = ob.OracleScene(3, origin, axes, flat=flat, params=params)
    o, cnt = osc.colors_at(xs.ravel(), ys.ravel(), w, h, counters=True)
    assert cnt["rays"] > 20 * w * h                       # most rays really go tens of levels deep
    for force_var in (False, True):
        if force_var:
            monkeypatch.setenv("NTRACER_FORCE_VAR", "1")
        sc2 = tracern.CompositeScene.from_flat(3, flat)
        sc2.set_params_flat(params)
        sc2._set_camera_arrays(origin, axes)
        c = sc2.colors_at(xs.ravel(), ys.ravel(), w, h)
        assert np.abs(c - o).max() < TOL_ORACLE, force_var
        img = render_host(sc2, fmt_of(w, h, fx.RGBF32)).view(">f4").reshape(h, w, 3)
        assert np.abs(img - np.clip(o, 0, 1).reshape(h, w, 3)).max() < TOL_ORACLE, force_var


def test_five_dimensional_feature_scene_vs_reference(monkeypatch):
    """feature5_n5 (captured from the reference's tracer5: transparent and reflective simplices, Solids, lights, shadows,
    reflection) through composite_kernel_t<5, true>: the reference's colours on every sample (1e-4, north_star's tolerance),
    the default-mode oracle's to 1e-5; with NTRACER_CLEAN_NORMALS=1 the clean-mode oracle's."""
    g = fx.load("feature5_n5")
    flat = fx.flat_of(g)
    p = fx.params_of(g)
    sc = tracern.CompositeScene.from_flat(5, flat)
    sc.set_params_flat(p)
    for clean in (False, True):
        if clean:
            monkeypatch.setenv("NTRACER_CLEAN_NORMALS", "1")
        for k, f in enumerate(g["frames"]):
            sc._set_camera_arrays(g["origins"][f], g["axes"][f])
            c = sc.colors_at(g["xs"], g["ys"], 160, 100)
            o = ob.OracleScene(5, g["origins"][f], g["axes"][f], flat=flat, params=p, clean_normals=clean).colors_at(g["xs"], g["ys"], 160, 100)
            assert np.abs(c - o).max() < TOL_ORACLE, (int(f), clean)
            if not clean:
                assert np.abs(c - g["colors"][k]).max() < TOL_REF, int(f)
    monkeypatch.delenv("NTRACER_CLEAN_NORMALS")
    sc._set_camera_arrays(g["origins"][9], g["axes"][9])
    img = render_host(sc, fmt_of(160, 100, fx.RGBF32))
    ref = ob.OracleScene(5, g["origins"][9], g["axes"][9], flat=flat, params=p).render(160, 100, fx.RGBF32, threads=7)
    assert np.abs(img.view(">f4") - ref.view(">f4")).max() < TOL_ORACLE


def test_shadow_rays_are_counted():
    g = fx.load("feature3d")
    flat = fx.flat_of(g, opaque=True)
    sc = tracern.CompositeScene.from_flat(3, flat)
    sc._set_camera_arrays(g["origin"], g["axes"])
    sc.set_params_flat(fx.params_of(g, "shadows__"))
    render_host(sc, fmt_of(96, 64, fx.RGBX8), collect_stats=True)
    st = sc.last_stats()
    assert st["shadow_rays"] > 0 and st["rays"] > 96 * 64      # primary + reflection


@pytest.mark.parametrize("name", ["cell120_n4", "orthoplex5_n5", "simplex10_n10", "feature5_n5", "lit12_n12", "feature16_n16"])
def test_composite_soak_slice(name):
    """A slice of tools/composite_soak.py in the suite: a golden scene as captured, lit with shadows (one light outside the scene's
    box, one inside it -- the shadow walk's far-child rule both ways -- and a global one) and with every material 30 % reflective,
    and lit under the native builder's k-d tree, four random cameras each, through the drop-in call against the oracle's frames: colours within 1e-5, and the cameras do see
    the scene.  The fixed-n packet and shading kernels (n = 4, 5, 10), the transparency kernels, and the run-time-n ones (12, 16)."""
    import bench
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import composite_soak
    threads = max(1, min(64, bench.cpu_quota_cores() - 1))
    res = composite_soak.soak_scene(name, 4, 991, 240, 150, threads)
    assert [r[0] for r in res] == ["captured", "lit", "mirror", "rebuilt"]
    for vname, n, worst, nbad, frames_bad, hits, shadow in res:
        assert nbad == 0 and worst < TOL_ORACLE, (name, vname, worst, nbad)
        assert hits > 0.003 * 4 * 240 * 150, (name, vname, hits)
        assert (shadow > 0) == (vname != "captured" or name in ("feature5_n5", "lit12_n12", "feature16_n16")), (name, vname, shadow)


@pytest.mark.parametrize("n", [3, 6, 8, 10, 19])
def test_box_soak_slice(n):
    """A slice of tools/box_soak.py in the suite: 24 random cameras per dimension at full 1920x1080 -- orthonormal frames at
    distances 1.2 .. 12 from the cube, off-axis, every seventh on a diagonal (coordinates equal up to rounding, like the demo's
    camera path: whole regions of near-ties between faces) -- one multi-frame launch per format, RGBX8 and three fp32
    channels, every frame byte for byte against the oracle.  This is what exercises the guard arithmetic of the tile
    kernel's lean loops (nt_box.hpp: the rsq quantisation's 2^-18 guard, the stretch codes' margins) beyond the 160 scripted
    cameras."""
    import torch
    import bench
    W, H, F = 1920, 1080, 24
    threads = max(1, min(64, bench.cpu_quota_cores() - 1))
    rng = np.random.default_rng(20260 + n)
    origins, axes = [], []
    for k in range(F):
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        q = np.ascontiguousarray(q, np.float32)
        dist = float(rng.choice([1.2, 1.6, 2.5, 4.0, 7.0, 12.0]))
        o = -q[2] * np.float32(dist) + np.float32(rng.uniform(-0.6, 0.6)) * q[0] + np.float32(rng.uniform(-0.6, 0.6)) * q[1]
        if k % 7 == 0:
            o = np.full(n, -dist / np.sqrt(n), np.float32)
            q = q.copy()
            q[2] = -o / np.linalg.norm(o)
        origins.append(o.astype(np.float32))
        axes.append(q)
    o = np.ascontiguousarray(np.stack(origins), np.float32)
    a = np.ascontiguousarray(np.stack(axes), np.float32)
    sc = tracern.BoxScene(n)
    osc = ob.OracleScene(n, o[0], a[0])
    for chans in (fx.RGBX8, fx.RGBF32):
        fmt = fmt_of(W, H, chans)
        fst = fmt._as_struct()
        fb = torch.zeros((F, H * fmt.pitch), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), H * fmt.pitch, F, o.ctypes.data_as(_lib.f32p),
                                                      a.ctypes.data_as(_lib.f32p), C.byref(fst), None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        got = fb.cpu().numpy().reshape(F, H, fmt.pitch)
        for f in range(F):
            osc.set_camera(o[f], a[f])
            ref = osc.render(W, H, chans, threads=threads)
            assert np.array_equal(got[f], ref), (n, f, fmt.bytes_per_pixel, int((got[f] != ref).sum()))


def test_camera_table_renders_the_same_frames():
    """nt_camera_table_create / nt_render_table_device (cameras of a path resident in device memory, one launch a call) against
    nt_render_frames_device (cameras packed and uploaded per call): the same bytes, for BoxScene -- whole frames and one
    rank's bands -- and for the 120-cell's packet kernel; parts of a table; and the refusals (wrong dimension, frames outside the table)."""
    import torch
    from ntracer_amd.render import CameraTable
    g = fx.load("box_n6_1920x1080")
    F = 12
    o = np.ascontiguousarray(g["origins"][10:10 + F], np.float32)
    a = np.ascontiguousarray(g["axes"][10:10 + F], np.float32)
    sc = tracern.BoxScene(6)
    fmt = fmt_of(1920, 1080, fx.RGBX8)
    fst = fmt._as_struct()
    tab = CameraTable(6, o, a)
    assert tab.frames == F
    st = torch.cuda.current_stream().cuda_stream
    for world, rank, brows in ((1, 0, 32), (8, 3, 8)):
        opts = _lib.NtRenderOpts()
        opts.device = -1
        opts.band_rank, opts.band_world, opts.band_rows, opts.compact = rank, world, brows, 1
        rows = len(ntd.owned_rows(1080, rank, world, brows))
        a_fb = torch.zeros((F, rows * fmt.pitch), dtype=torch.uint8, device="cuda")
        b_fb = torch.zeros_like(a_fb)
        _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(a_fb.data_ptr()), rows * fmt.pitch, F, o.ctypes.data_as(_lib.f32p),
                                                      a.ctypes.data_as(_lib.f32p), C.byref(fst), C.byref(opts), C.c_void_p(st)))
        assert tab.render(sc, b_fb, fmt, band_rank=rank, band_world=world, compact=True, band_rows=brows)
        torch.cuda.synchronize()
        assert torch.equal(a_fb, b_fb), (world, rank)
    ref = ob.OracleScene(6, o[5], a[5]).render(1920, 1080, fx.RGBX8, threads=8)
    full = torch.zeros((F, 1080 * fmt.pitch), dtype=torch.uint8, device="cuda")
    tab.render(sc, full, fmt)
    torch.cuda.synchronize()
    assert np.array_equal(full[5].cpu().numpy().reshape(1080, fmt.pitch), ref)
    with pytest.raises(ValueError, match="dimensions"):
        tab.render(tracern.BoxScene(5), full, fmt)
    # part of a table: frames [first, first + count) -- also of BoxScene(10), whose second kernel reads the cameras too
    part = torch.zeros((4, 1080 * fmt.pitch), dtype=torch.uint8, device="cuda")
    assert tab.render(sc, part, fmt, first=2, count=4)
    torch.cuda.synchronize()
    assert torch.equal(part, full[2:6])
    for first, count in ((-1, 2), (F - 1, 2), (0, 0), (0, F + 1)):
        with pytest.raises(ValueError):
            _lib.check(_lib.lib().nt_render_table_device(sc._handle, C.c_void_p(full.data_ptr()), 1080 * fmt.pitch, tab._h, first, count, C.byref(fst), None,
                                                         C.c_void_p(st)))
    g10 = fx.load("box_n10_4096x4096")
    o10 = np.ascontiguousarray(g10["origins"][[5, 60, 61, 130, 140]], np.float32)
    a10 = np.ascontiguousarray(g10["axes"][[5, 60, 61, 130, 140]], np.float32)
    sc10 = tracern.BoxScene(10)
    fmt10 = fmt_of(1024, 640, fx.RGBX8)
    tab10 = CameraTable(10, o10, a10)
    whole10 = torch.zeros((5, 640 * fmt10.pitch), dtype=torch.uint8, device="cuda")
    part10 = torch.zeros((2, 640 * fmt10.pitch), dtype=torch.uint8, device="cuda")
    tab10.render(sc10, whole10, fmt10)
    tab10.render(sc10, part10, fmt10, first=3, count=2)
    torch.cuda.synchronize()
    assert torch.equal(part10, whole10[3:5])
    assert np.array_equal(part10[0].cpu().numpy().reshape(640, fmt10.pitch), ob.OracleScene(10, o10[3], a10[3]).render(1024, 640, fx.RGBX8, threads=8))
    g4 = fx.load("cell120_n4")
    sc4 = tracern.CompositeScene.from_flat(4, fx.flat_of(g4))
    o4 = np.ascontiguousarray(g4["origins"][[0, 40, 93]], np.float32)
    a4 = np.ascontiguousarray(g4["axes"][[0, 40, 93]], np.float32)
    fmt4 = fmt_of(640, 360, fx.RGBX8)
    fst4 = fmt4._as_struct()
    x = torch.zeros((3, 360 * fmt4.pitch), dtype=torch.uint8, device="cuda")
    y = torch.zeros_like(x)
    _lib.check(_lib.lib().nt_render_frames_device(sc4._handle, C.c_void_p(x.data_ptr()), 360 * fmt4.pitch, 3, o4.ctypes.data_as(_lib.f32p),
                                                  a4.ctypes.data_as(_lib.f32p), C.byref(fst4), None, C.c_void_p(st)))
    CameraTable(4, o4, a4).render(sc4, y, fmt4)
    torch.cuda.synchronize()
    assert torch.equal(x, y)


def test_box_kernel_paths_alternate_on_one_scene(monkeypatch):
    """The packed-RGB formats normally take the fused tile kernel; NTRACER_BOX_PATH=0 selects the older cull -> box -> redo
    kernels, which share the scene's scratch buffer with it (the fused path keeps a bitmap there that must be all zero between
    launches).  The same frames on ONE scene through path 0, the fused path and path 0 again -- and the fused path with rows
    interleaved and not -- byte for byte like the oracle, in 6 and in 10 dimensions (with and without a second kernel)."""
    import torch
    for n, name, W, H in ((6, "box_n6_1920x1080", 1920, 1080), (10, "box_n10_4096x4096", 1024, 640)):
        g = fx.load(name)
        sel = [3, 47, 101]
        o = np.ascontiguousarray(g["origins"][sel], np.float32)
        a = np.ascontiguousarray(g["axes"][sel], np.float32)
        fmt = fmt_of(W, H, fx.RGBX8)
        fst = fmt._as_struct()
        refs = [ob.OracleScene(n, o[k], a[k]).render(W, H, fx.RGBX8, threads=8) for k in range(len(sel))]
        sc = tracern.BoxScene(n)
        fb = torch.empty((len(sel), fmt.pitch * H), dtype=torch.uint8, device="cuda")
        for path, il in (("0", None), (None, None), ("0", None), (None, "0"), (None, None), ("0", None)):
            if path is None:
                monkeypatch.delenv("NTRACER_BOX_PATH", raising=False)
            else:
                monkeypatch.setenv("NTRACER_BOX_PATH", path)
            if il is None:
                monkeypatch.delenv("NTRACER_BOX_INTERLEAVE", raising=False)
            else:
                monkeypatch.setenv("NTRACER_BOX_INTERLEAVE", il)
            fb.fill_(0x5a)
            _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), fmt.pitch * H, len(sel), o.ctypes.data_as(_lib.f32p),
                                                          a.ctypes.data_as(_lib.f32p), C.byref(fst), None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            got = fb.cpu().numpy().reshape(len(sel), H, fmt.pitch)
            for k in range(len(sel)):
                assert np.array_equal(got[k], refs[k]), (n, path, il, k, int((got[k] != refs[k]).sum()))


def test_render_calls_captured_in_a_hip_graph():
    """After one warm-up call on the capture stream (scratch allocated, row table cached, no change of stream to drain) a
    render call from a camera table is nothing but kernel launches and can be captured into a HIP graph (torch.cuda.graph)
    and replayed: BoxScene(6) (one kernel a call) and BoxScene(10) (tile kernel + second kernel) -- the replay's bytes are the
    direct call's.  nt_render_frames_device, which stages the caller's host cameras on every call, refuses to be captured."""
    import torch
    from ntracer_amd.render import CameraTable
    s = torch.cuda.Stream()
    L = _lib.lib()
    for n, name, W, H in ((6, "box_n6_1920x1080", 1920, 1080), (10, "box_n10_4096x4096", 1024, 640)):
        g = fx.load(name)
        F = 6
        o = np.ascontiguousarray(g["origins"][20:20 + F], np.float32)
        a = np.ascontiguousarray(g["axes"][20:20 + F], np.float32)
        fmt = fmt_of(W, H, fx.RGBX8)
        fst = fmt._as_struct()
        sc = tracern.BoxScene(n)
        tab = CameraTable(n, o, a)
        ref = torch.zeros((F, H * fmt.pitch), dtype=torch.uint8, device="cuda")
        fb = torch.zeros_like(ref)

        def call(buf):
            return L.nt_render_table_device(sc._handle, C.c_void_p(buf.data_ptr()), H * fmt.pitch, tab._h, 0, F, C.byref(fst), None, C.c_void_p(s.cuda_stream))
        with torch.cuda.stream(s):
            _lib.check(call(ref))
        s.synchronize()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        refused = None
        with torch.cuda.graph(gr, stream=s):
            _lib.check(call(fb))
            _lib.check(call(fb))
            if n == 6:
                refused = L.nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), H * fmt.pitch, F, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p),
                                                    C.byref(fst), None, C.c_void_p(s.cuda_stream))
        assert refused is None or refused == _lib.NT_E_UNSUPPORTED
        for rep in range(2):
            fb.zero_()
            torch.cuda.synchronize()
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(fb, ref), (n, rep)
        del gr
        assert np.array_equal(ref[2].cpu().numpy().reshape(H, fmt.pitch), ob.OracleScene(n, o[2], a[2]).render(W, H, fx.RGBX8, threads=8))


def test_overlapped_hint_changes_the_launch_shape_not_the_bytes():
    """nt_render_opts.overlapped (the caller keeps several streams busy: long waves from 64 rows up) must never change a pixel:
    a rank's eighth of 160 headline frames -- 136 rows in 8-row bands, the launch whose shape the hint changes -- and the whole
    frames, with the hint and without, alternating on one scene, on two streams at once; three frames against the oracle."""
    import torch
    from ntracer_amd import distributed as ntd
    g = fx.load("box_n6_1920x1080")
    W, H, F = 1920, 1080, 160
    o = np.ascontiguousarray(g["origins"][:F], np.float32)
    a = np.ascontiguousarray(g["axes"][:F], np.float32)
    fmt = fmt_of(W, H, fx.RGBX8)
    fst = fmt._as_struct()
    L = _lib.lib()
    for world, brows in ((8, 8), (1, 32)):
        own = ntd.owned_rows(H, 0, world, brows)
        scenes = [tracern.BoxScene(6), tracern.BoxScene(6)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        fbs = [torch.empty((F, len(own) * fmt.pitch), dtype=torch.uint8, device="cuda") for _ in range(2)]
        got = {}
        for hint in (0, 1, 0, 1):
            opts = _lib.NtRenderOpts()
            opts.device = -1
            opts.band_rank, opts.band_world, opts.band_rows, opts.compact = 0, world, brows, 1
            opts.overlapped = hint
            for fb in fbs:
                fb.fill_(0x5a)
            torch.cuda.synchronize()
            for k in range(4):                           # consecutive calls overlap: two handles, two buffers, two streams
                _lib.check(L.nt_render_frames_device(scenes[k & 1]._handle, C.c_void_p(fbs[k & 1].data_ptr()), len(own) * fmt.pitch, F,
                                                     o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(fst), C.byref(opts),
                                                     C.c_void_p(streams[k & 1].cuda_stream)))
            torch.cuda.synchronize()
            assert torch.equal(fbs[0], fbs[1])
            if hint in got:
                assert torch.equal(got[hint], fbs[0])
            got[hint] = fbs[0].clone()
        assert torch.equal(got[0], got[1]), (world, int((got[0] != got[1]).sum()))
        for k in (0, 77, 159):
            ref = ob.OracleScene(6, o[k], a[k]).render(W, H, fx.RGBX8, threads=8)
            assert np.array_equal(got[1][k].cpu().numpy().reshape(len(own), fmt.pitch), ref[own]), (world, k)


def test_asking_for_statistics_does_not_change_the_pixels():
    """collect_stats on a scene with Solids: the frame still comes from the kernel that reproduces the reference's normal
    handling (the counters from a launch of their own); on a scene with transparent materials -- whose kernels keep no
    counters -- it is refused instead of reporting nothing."""
    g = fx.load("feature3d")
    flat = fx.flat_of(g, opaque=True)
    sc = tracern.CompositeScene.from_flat(3, flat)
    sc._set_camera_arrays(g["origin"], g["axes"])
    sc.set_params_flat(fx.params_of(g, "shadows__"))
    fmt = fmt_of(96, 64, fx.RGBF32)
    plain = render_host(sc, fmt)
    counted = render_host(sc, fmt, collect_stats=True)
    assert np.array_equal(plain, counted)
    st = sc.last_stats()
    assert st["rays"] > 96 * 64 and st["solid_tests"] > 0
    tr = tracern.CompositeScene.from_flat(3, fx.flat_of(g))            # the fixture as it is: transparent materials
    tr._set_camera_arrays(g["origin"], g["axes"])
    with pytest.raises(NotImplementedError, match="collect_stats"):
        render_host(tr, fmt, collect_stats=True)
    assert not tr.locked


# ------------------------------------------------------------------ renderer protocol
def _long_scene():
    """the 120-cell walked strictly at 4096x4096: eight slabs of a few milliseconds each"""
    g = fx.load("cell120_n4")
    sc = tracern.CompositeScene.from_flat(4, fx.flat_of(g))
    sc._set_camera_arrays(g["origins"][0], g["axes"][0])
    fmt = fmt_of(4096, 4096, fx.RGBX8)
    render_host(sc, fmt_of(64, 64, fx.RGBX8))            # scene upload, code load: not part of what is timed below
    return sc, fmt


def test_abort_flag_set_before_the_call_renders_nothing():
    """signal_abort / renderer::CANCEL (render.cpp:911-923, polled at :412): a flag that is already up stops the render
    before its first slab; the caller's buffer is returned untouched and the scene is unlocked."""
    sc, fmt = _long_scene()
    buf = bytearray(b"\xab" * (fmt.pitch * fmt.height))
    arr = (C.c_char * len(buf)).from_buffer(buf)
    flag = C.c_int(1)
    fst = fmt._as_struct()
    r = _lib.lib().nt_render(sc._handle, arr, len(buf), C.byref(fst), None, C.byref(flag))
    assert r == _lib.NT_ABORTED
    assert bytes(buf) == b"\xab" * len(buf)
    assert not sc.locked


def test_signal_abort_stops_a_running_render():
    """The flag goes up from another thread as soon as the render holds the scene: render() returns False well before a full
    frame's time has passed (the kernels read the abort word when a block starts: what is in flight finishes, the rest leaves),
    nothing of the incomplete frame is copied back -- the caller's buffer is as it was -- and the scene is unlocked again."""
    import threading
    import time
    sc, fmt = _long_scene()
    H, pitch = fmt.height, fmt.pitch
    r = ntracer_amd.BlockingRenderer()
    full = bytearray(pitch * H)
    assert r.render(full, fmt, sc, strict_reference=True) is True         # (first use of this size: allocations)
    t0 = time.perf_counter()
    assert r.render(full, fmt, sc, strict_reference=True) is True
    t_full = time.perf_counter() - t0

    buf = bytearray(b"\xab" * (pitch * H))
    out = {}

    def run():
        out["ok"] = r.render(buf, fmt, sc, strict_reference=True)
        out["end"] = time.perf_counter()

    t = threading.Thread(target=run)
    t.start()
    while not sc.locked and t.is_alive():
        pass
    time.sleep(0.3 * t_full)                     # somewhere in the middle of the frame
    t_sig = time.perf_counter()
    r.signal_abort()
    t.join()
    assert out["ok"] is False
    assert not sc.locked
    assert bytes(buf) == b"\xab" * len(buf)
    # after the flag went up: the waves in flight -- well under half a frame
    assert out["end"] - t_sig < 0.5 * t_full, (out["end"] - t_sig, t_full)
    assert r.render(buf, fmt, sc, strict_reference=True) is True        # state is reset at the start of the next render (render.cpp:889)
    assert bytes(buf) == bytes(full)


def test_abort_word_on_the_batched_device_path():
    """nt_render_opts.abort_device: the device entry points only enqueue, so their abort is a dword the device can read while
    the kernels run (here: device memory).  Raised before the call: no block draws -- every frame keeps its sentinel bytes;
    lowered again: the same call renders the frames."""
    import torch
    g = fx.load("box_n6_1920x1080")
    sc = tracern.BoxScene(6)
    fmt = fmt_of(1920, 1080, fx.RGBX8)
    fst = fmt._as_struct()
    F = 8
    word = torch.zeros(16, dtype=torch.int32, device="cuda")
    opts = _lib.NtRenderOpts()
    opts.device = -1
    opts.band_world = 1
    opts.abort_device = word.data_ptr()
    o = np.ascontiguousarray(g["origins"][:F], np.float32)
    a = np.ascontiguousarray(g["axes"][:F], np.float32)
    fb = torch.full((F, fmt.pitch * 1080), 0xab, dtype=torch.uint8, device="cuda")

    def go():
        _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), fmt.pitch * 1080, F, o.ctypes.data_as(_lib.f32p),
                                                      a.ctypes.data_as(_lib.f32p), C.byref(fst), C.byref(opts),
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
    word.fill_(1)
    torch.cuda.synchronize()
    go()
    assert bool((fb == 0xab).all())
    word.fill_(0)
    torch.cuda.synchronize()
    go()
    ref = ob.OracleScene(6, o[3], a[3]).render(1920, 1080, fx.RGBX8, threads=8)
    assert np.array_equal(fb[3].cpu().numpy().reshape(1080, fmt.pitch), ref)
    # ... and the 120-cell's packet kernel the same way
    g4 = fx.load("cell120_n4")
    sc4 = tracern.CompositeScene.from_flat(4, fx.flat_of(g4))
    o4 = np.ascontiguousarray(g4["origins"][:2], np.float32)
    a4 = np.ascontiguousarray(g4["axes"][:2], np.float32)
    fmt4 = fmt_of(640, 360, fx.RGBX8)
    fst4 = fmt4._as_struct()
    fb4 = torch.full((2, fmt4.pitch * 360), 0xab, dtype=torch.uint8, device="cuda")
    word.fill_(1)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().nt_render_frames_device(sc4._handle, C.c_void_p(fb4.data_ptr()), fmt4.pitch * 360, 2, o4.ctypes.data_as(_lib.f32p),
                                                  a4.ctypes.data_as(_lib.f32p), C.byref(fst4), C.byref(opts), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert bool((fb4 == 0xab).all())


def test_second_render_on_a_busy_renderer_or_scene_is_refused():
    """already_running_error (render.cpp:87-92, 881): a renderer refuses a second render() while one is running; so does
    the scene handle for a second renderer (NT_E_BUSY), and its mutators raise LockedError (ntracer_body.hpp:235-240)."""
    import threading
    sc, fmt = _long_scene()
    r = ntracer_amd.BlockingRenderer()
    buf = bytearray(fmt.pitch * fmt.height)
    other = bytearray(fmt.pitch * fmt.height)
    out = {}

    def run():
        out["ok"] = r.render(buf, fmt, sc, strict_reference=True)

    t = threading.Thread(target=run)
    t.start()
    while not sc.locked and t.is_alive():
        pass
    try:
        assert t.is_alive()
        with pytest.raises(RuntimeError, match="already running"):
            r.render(other, fmt, sc)
        with pytest.raises(RuntimeError, match="already running"):
            ntracer_amd.BlockingRenderer().render(other, fmt, sc)
        with pytest.raises(_lib.LockedError):
            sc.set_fov(0.5)
    finally:
        r.signal_abort()
        t.join()
    assert not sc.locked
    sc.set_fov(0.8)
    assert r.render(buf, fmt, sc) is True


def test_callback_renderer_invokes_callback_and_locks_scene():
    import threading
    sc = tracern.BoxScene(4)
    fmt = fmt_of(320, 200, fx.RGBX8)
    buf = bytearray(fmt.pitch * 200)
    done = threading.Event()
    r = ntracer_amd.CallbackRenderer()
    r.begin_render(buf, fmt, sc, lambda rr: done.set())
    assert done.wait(30)
    r.abort_render()
    assert not sc.locked
    assert any(buf)

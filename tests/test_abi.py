"""The C-ABI library: loads, exports every symbol include/ntracer_hip.h declares, validates its inputs
like the reference, and -- without a GPU -- fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import fixtures as fx
import ntracer_amd
from ntracer_amd import _lib, tracern

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "ntracer_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 20
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libntracer_hip.so does not export %s" % n
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == names     # the binding covers the whole header


def test_version_and_error_channel():
    L = _lib.lib()
    assert b"gfx950" in L.nt_version()
    assert L.nt_box_scene_create(2) is None
    assert "dimension" in _lib.last_error()


def test_scene_state_and_locking():
    s = tracern.BoxScene(5)
    assert s.dimension == 5 and abs(s.fov - 0.8) < 1e-7 and not s.locked
    cam = tracern.Camera(5)
    cam.translate(tracern.Vector.axis(5, 2, -5))
    s.set_camera(cam)
    assert list(s.get_camera().origin) == [0, 0, -5, 0, 0]
    L = _lib.lib()
    assert L.nt_scene_lock(s._handle) == 0
    assert s.locked
    with pytest.raises(ntracer_amd.LockedError):
        s.set_camera(cam)
    with pytest.raises(ntracer_amd.LockedError):
        s.set_fov(1.0)
    assert L.nt_scene_unlock(s._handle) == 0
    s.set_fov(1.0)
    assert abs(s.fov - 1.0) < 1e-7
    assert L.nt_scene_unlock(s._handle) == _lib.NT_E_INVALID
    with pytest.raises(TypeError):
        s.set_camera(tracern.Camera(4))


def test_format_validation_matches_reference_messages():
    Ch, IF = ntracer_amd.Channel, ntracer_amd.ImageFormat
    with pytest.raises(ValueError, match="can only be 32"):
        Ch(16, 1, 0, 0, 0, True)
    with pytest.raises(ValueError, match="cannot be greater than 31"):
        Ch(32, 1, 0, 0)
    with pytest.raises(ValueError, match="cannot be less than 1"):
        Ch(0, 1, 0, 0)
    with pytest.raises(ValueError, match="Too many bytes per pixel"):
        IF(4, 4, [Ch(31, 1, 0, 0)] * 5)
    with pytest.raises(ValueError, match="at least 1"):
        IF(0, 4, [Ch(8, 1, 0, 0)])
    with pytest.raises(ValueError, match="pitch"):
        IF(10, 4, [Ch(8, 1, 0, 0)], pitch=9)
    f = IF(10, 4, [Ch(5, 1, 0, 0), Ch(6, 0, 1, 0), Ch(5, 0, 0, 1)])
    assert f.bytes_per_pixel == 2 and f.pitch == 20
    # the C side applies the same rules (a C++ host gets the same errors)
    st = f._as_struct()
    assert _lib.lib().nt_format_bytes_per_pixel(C.byref(st)) == 2
    st.pitch = 5
    assert _lib.lib().nt_format_bytes_per_pixel(C.byref(st)) == _lib.NT_E_INVALID


def test_scene_description_is_validated():
    g = fx.load("cell600_n4")
    flat = fx.flat_of(g)
    sc = tracern.CompositeScene.from_flat(4, flat)
    assert sc.dimension == 4 and sc.max_reflect_depth == 4 and sc.camera_light and not sc.shadows
    bad = dict(flat)
    bad["node_left"] = np.array(flat["node_left"]).copy()
    bad["node_left"][0] = 10 ** 6                        # child out of range
    with pytest.raises(ValueError, match="out of range|out of bounds"):
        tracern.CompositeScene.from_flat(4, bad)
    bad = dict(flat)
    bad["items"] = np.array(flat["items"]).copy()
    bad["items"][3] = (10 ** 6) << 2                     # primitive out of range
    with pytest.raises(ValueError, match="out of range"):
        tracern.CompositeScene.from_flat(4, bad)
    bad = dict(flat)
    nl = np.array(flat["node_left"]).copy()
    ax = np.array(flat["node_axis"])
    branch = int(np.nonzero(ax >= 0)[0][1])
    nl[branch] = 0                                       # cycle back to the root
    bad["node_left"] = nl
    with pytest.raises(ValueError, match="more than once"):
        tracern.CompositeScene.from_flat(4, bad)


@pytest.mark.skipif(_lib.lib().nt_device_count() > 0, reason="checks the no-GPU behaviour")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    s = tracern.BoxScene(3)
    fmt = ntracer_amd.ImageFormat(8, 8, [ntracer_amd.Channel(8, 1, 0, 0)])
    with pytest.raises(RuntimeError, match="no HIP device"):
        ntracer_amd.BlockingRenderer().render(bytearray(64), fmt, s)
    with pytest.raises(RuntimeError, match="no HIP device"):
        s.calculate_color(1, 1, 8, 8)
    assert not s.locked          # the failed render released the scene


def test_render_argument_checks_come_before_the_device():
    s = tracern.BoxScene(3)
    fmt = ntracer_amd.ImageFormat(8, 8, [ntracer_amd.Channel(8, 1, 0, 0)])
    with pytest.raises(ValueError, match="too small"):
        ntracer_amd.BlockingRenderer().render(bytearray(63), fmt, s)
    with pytest.raises((BufferError, TypeError)):
        ntracer_amd.BlockingRenderer().render(bytes(64), fmt, s)
    with pytest.raises(TypeError):
        ntracer_amd.BlockingRenderer().render(bytearray(64), fmt, object())


def test_limits_are_refused_not_approximated():
    """What the library cannot do it refuses: dimensions outside 3..64, records that do not match the dimension.  (Round 2:
    transparency at any n and any max_reflect_depth are no longer among them -- tests/test_gpu_parity.py.)"""
    with pytest.raises(ValueError, match="dimension"):
        tracern.BoxScene(65)
    with pytest.raises(ValueError, match="dimension"):
        tracern.BoxScene(2)
    flat = fx.flat_of(fx.load("cell600_n4"))
    with pytest.raises(ValueError):
        tracern.CompositeScene.from_flat(9, flat)                 # record sizes do not match dimension 9


def test_header_is_c99_and_links_from_plain_c(tmp_path):
    """include/ntracer_hip.h must be usable from C (the boundary is a C ABI): gcc -std=c99 compiles tests/c_client.c
    against it, links libntracer_hip.so, and the program talks to the library."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    exe = str(tmp_path / "c_client")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_client.c"), "-o", exe, "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH),
                           "-Wl,-rpath," + libdir])
    out = subprocess.check_output([exe]).decode()
    assert "dimension 6 composite 0" in out and "kdtree: status 0 nodes 1 items 2" in out

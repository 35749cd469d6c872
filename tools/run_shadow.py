#!/usr/bin/env python3
"""The 120-cell with one point light, one global light and shadows on (SURVEY 8d's "+shadow" variant of config 4) -- target for
rocprofv3 runs.   python3 tools/run_shadow.py [width height [frames]]      SHADOWS=0: lights without shadow rays"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntracer_amd  # noqa: E402
from ntracer_amd import _lib, tracern  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
F = int(sys.argv[3]) if len(sys.argv) > 3 else 8
g = np.load(os.path.join(ROOT, "tests", "golden", "cell120_n4.npz"))
sc = tracern.CompositeScene.from_flat(4, g)
sc.add_light(tracern.PointLight(tracern.Vector(4, (8.0, 9.0, -7.0, 3.0)), (60.0, 60.0, 60.0)))
sc.add_light(tracern.GlobalLight(tracern.Vector(4, (0.2, -1.0, 0.3, 0.1)).unit(), (0.5, 0.5, 0.5)))
sc.set_shadows(os.environ.get("SHADOWS", "1") != "0")
chan = [ntracer_amd.Channel(8, 1, 0, 0), ntracer_amd.Channel(8, 0, 1, 0), ntracer_amd.Channel(8, 0, 0, 1), ntracer_amd.Channel(8, 0, 0, 0)]
fmt = ntracer_amd.ImageFormat(W, H, chan)
fst = fmt._as_struct()
sel = [(i * 160) // F for i in range(F)]
o = np.ascontiguousarray(g["origins"][sel], np.float32)
a = np.ascontiguousarray(g["axes"][sel], np.float32)
# rays of frame 0 (a statistics render: counts shadow rays with device atomics)
sc._set_camera_arrays(g["origins"][0], g["axes"][0])
buf = bytearray(fmt.pitch * H)
ntracer_amd.BlockingRenderer().render(buf, fmt, sc, collect_stats=True)
stats = sc.last_stats()
fb = torch.empty((F, fmt.pitch * H), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
for rep in range(3):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), fmt.pitch * H, F, o.ctypes.data_as(_lib.f32p),
                                                  a.ctypes.data_as(_lib.f32p), C.byref(fst), None, C.c_void_p(st.cuda_stream)))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / F
    print("shadow scene %dx%d, %d frames a call: %.3f ms/frame; frame 0 casts %d primary + %d shadow rays -> %.1f Mrays/s primary+shadow"
          % (W, H, F, ms, stats["rays"], stats["shadow_rays"], (stats["rays"] + stats["shadow_rays"]) / ms / 1e3))
print("checksum", int(fb.to(torch.int64).sum().item()))

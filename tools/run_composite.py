#!/usr/bin/env python3
"""Render the 120-cell (config 4) a few times -- target for rocprofv3 runs.
usage: run_composite.py [fixture [frames [rebuilt]]]   ("rebuilt": the same primitives under our own k-d tree)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntracer_amd  # noqa: E402
from ntracer_amd import _lib, tracern  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cell120_n4"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
n = int(g["dimension"])
sc = tracern.CompositeScene.from_flat(n, g)
if len(sys.argv) > 3 and sys.argv[3] == "rebuilt":
    sc = sc.with_rebuilt_tree()
W, H = 1920, 1080
fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(8, 1, 0, 0), ntracer_amd.Channel(8, 0, 1, 0),
                                     ntracer_amd.Channel(8, 0, 0, 1), ntracer_amd.Channel(8, 0, 0, 0)])
fst = fmt._as_struct()
sel = [(i * 160) // frames for i in range(frames)]
o = np.ascontiguousarray(g["origins"][sel], np.float32)
a = np.ascontiguousarray(g["axes"][sel], np.float32)
fb = torch.empty((frames, fmt.pitch * H), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
for rep in range(2):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), fmt.pitch * H, frames,
                                                  o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(fst), None,
                                                  C.c_void_p(st.cuda_stream)))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / frames
    print("%s: %.3f ms/frame, %.1f Mrays/s" % (name, ms, W * H / ms / 1e3))
print("checksum", int(fb.to(torch.int64).sum().item()))

#!/usr/bin/env python3
"""How long does the chip take to settle under the headline workload?  Back-to-back 160-frame calls for `seconds`, the time of
every block of `block` calls (HIP events).   python3 tools/settle_probe.py [seconds [block]]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntracer_amd  # noqa: E402
from ntracer_amd import _lib, tracern  # noqa: E402
import bench  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
block = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = np.load(os.path.join(ROOT, "tests", "golden", "box_n6_1920x1080.npz"))
o = np.ascontiguousarray(g["origins"], np.float32)
a = np.ascontiguousarray(g["axes"], np.float32)
F, W, H = 160, 1920, 1080
fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in bench.RGBX8])
fst = fmt._as_struct()
fb = torch.empty((F, fmt.pitch * H), dtype=torch.uint8, device="cuda")
sc = tracern.BoxScene(6)
st = torch.cuda.current_stream()
L = _lib.lib()
args = (sc._handle, C.c_void_p(fb.data_ptr()), fmt.pitch * H, F, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(fst), None,
        C.c_void_p(st.cuda_stream))
L.nt_render_frames_device(*args)
torch.cuda.synchronize()
time.sleep(0.5)                       # idle first, like a fresh process
evs = []
t0 = time.perf_counter()
e = torch.cuda.Event(enable_timing=True)
e.record(st)
evs.append(e)
while time.perf_counter() - t0 < seconds:
    for _ in range(block):
        L.nt_render_frames_device(*args)
    e = torch.cuda.Event(enable_timing=True)
    e.record(st)
    evs.append(e)
    if len(evs) % 8 == 0:
        evs[-6].synchronize()          # keep the queue a few blocks deep, not unbounded
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) / block for i in range(len(evs) - 1)]
t = np.cumsum([0.0] + [m * block for m in ms])[:-1]
print("calls of 160 frames, %d per block; ms per call by time since the load began:" % block)
marks = [0, 2, 5, 10, 20, 50, 100, 200, 500, 1000, 2000, 3000, 5000, 8000]
for lo, hi in zip(marks[:-1], marks[1:]):
    sel = [m for m, tt in zip(ms, t) if lo <= tt < hi]
    if sel:
        print("  %5d .. %5d ms: %.4f ms a call (%d blocks) = %.0f Grays/s" % (lo, hi, float(np.mean(sel)), len(sel), W * H * F / np.mean(sel) / 1e6))

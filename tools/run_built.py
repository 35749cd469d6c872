#!/usr/bin/env python3
"""Scenes generated and partitioned on our side (ntracer_amd.polytope + the native k-d builder) against the scenes the
reference built (fixtures): 1080p frame time on the GPU, same cameras."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntracer_amd  # noqa: E402
from ntracer_amd import _lib, polytope, tracern  # noqa: E402

W, H = 1920, 1080
fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(8, 1, 0, 0), ntracer_amd.Channel(8, 0, 1, 0),
                                     ntracer_amd.Channel(8, 0, 0, 1), ntracer_amd.Channel(8, 0, 0, 0)])


def time_scene(sc, origins, axes, frames=8):
    fst = fmt._as_struct()
    sel = [(i * len(origins)) // frames for i in range(frames)]
    o = np.ascontiguousarray(origins[sel], np.float32)
    a = np.ascontiguousarray(axes[sel], np.float32)
    fb = torch.empty((frames, fmt.pitch * H), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), fmt.pitch * H, frames,
                                                      o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(fst), None,
                                                      C.c_void_p(st.cuda_stream)))
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / frames)
    return best, fb


for name, sym in (("cell600_n4", ["3", "3", "5"]), ("cell120_n4", ["5/2", "3", "3"])):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    ref = tracern.CompositeScene.from_flat(4, g)
    t = time.time()
    nt, ours, dist = polytope.build_scene(sym)
    tb = time.time() - t
    ms_ref, fb_ref = time_scene(ref, g["origins"], g["axes"])
    ms_ours, fb_ours = time_scene(ours, g["origins"], g["axes"])
    diff = (fb_ref != fb_ours).any(dim=1).sum().item(), ((fb_ref.to(torch.int16) - fb_ours.to(torch.int16)).abs() > 2).sum().item()
    print("%s: reference-built scene %.3f ms/frame; generated + our tree %.3f ms/frame (built in %.1f s); bytes differing by >2: %d of %d"
          % (name, ms_ref, ms_ours, tb, diff[1], fb_ref.numel()))

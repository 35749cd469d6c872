#!/usr/bin/env python3
"""One rank's share of an 8-GPU step (rank 0's bands of the 160 headline frames), issued on ONE stream and alternately on TWO
(two scene handles, two sets of framebuffers): does the tail of one call overlap the ramp of the next?
    python3 tools/two_stream_probe.py [world]"""
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
from ntracer_amd import distributed as ntd  # noqa: E402
import bench  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = np.load(os.path.join(ROOT, "tests", "golden", "box_n6_1920x1080.npz"))
o = np.ascontiguousarray(g["origins"], np.float32)
a = np.ascontiguousarray(g["axes"], np.float32)
F, W, H = 160, 1920, 1080
fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in bench.RGBX8])
fst = fmt._as_struct()
brows = bench.pick_band_rows(ntd, H, world)
opts = _lib.NtRenderOpts()
opts.device = 0
opts.band_rank, opts.band_world, opts.band_rows, opts.compact = 0, world, brows, 1
own = len(ntd.owned_rows(H, 0, world, brows))
L = _lib.lib()
scenes = [tracern.BoxScene(6), tracern.BoxScene(6)]
fbs = [torch.empty((F, own * fmt.pitch), dtype=torch.uint8, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
tabs = [L.nt_camera_table_create(6, F, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), 0) for _ in range(2)]


def call(i, table):
    if table:
        _lib.check(L.nt_render_table_device(scenes[i]._handle, C.c_void_p(fbs[i].data_ptr()), own * fmt.pitch, C.c_void_p(tabs[i]), 0, F, C.byref(fst), C.byref(opts),
                                            C.c_void_p(streams[i].cuda_stream)))
    else:
        _lib.check(L.nt_render_frames_device(scenes[i]._handle, C.c_void_p(fbs[i].data_ptr()), own * fmt.pitch, F, o.ctypes.data_as(_lib.f32p),
                                             a.ctypes.data_as(_lib.f32p), C.byref(fst), C.byref(opts), C.c_void_p(streams[i].cuda_stream)))


for table in (False, True):
    for two in (False, True):
        res = []
        for rep in range(3):
            t_end = time.perf_counter() + 0.25
            k = 0
            while time.perf_counter() < t_end:
                call(k % 2 if two else 0, table)
                k += 1
                if k % 8 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            steps = 200
            t0 = time.perf_counter()
            for k in range(steps):
                call(k % 2 if two else 0, table)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e6)
        print("world %d, %s, %s: %.1f us a step (best of 3: %s)" % (world, "camera table" if table else "host cameras", "two streams" if two else "one stream",
                                                                     min(res), ", ".join("%.1f" % r for r in res)))

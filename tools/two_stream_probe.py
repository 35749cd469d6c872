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
DIM = 6
F, W, H = 160, 1920, 1080
if os.environ.get("CASE") == "cfg5":                      # BASELINE configs[4]: BoxScene(10) 4096 x 4096, 16 frames a call
    DIM, F, W, H = 10, 16, 4096, 4096
g = np.load(os.path.join(ROOT, "tests", "golden", "box_n%d_%dx%d.npz" % (DIM, W, H)))
sel = (np.arange(F) * (len(g["origins"]) // F)) % len(g["origins"])
o = np.ascontiguousarray(g["origins"][sel], np.float32)
a = np.ascontiguousarray(g["axes"][sel], np.float32)
fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in bench.RGBX8])
fst = fmt._as_struct()
brows = bench.pick_band_rows(ntd, H, world)
opts = _lib.NtRenderOpts()
opts.device = 0
opts.band_rank, opts.band_world, opts.band_rows, opts.compact = 0, world, brows, 1
opts_ov = _lib.NtRenderOpts()
C.memmove(C.byref(opts_ov), C.byref(opts), C.sizeof(opts))
opts_ov.overlapped = int(os.environ.get("OVERLAPPED", "1"))          # what the several-stream legs tell the library
own = len(ntd.owned_rows(H, 0, world, brows))
L = _lib.lib()
# (other scenes of the process, each drawn once through the device entry point: do they cost the overlap?)
others = [tracern.BoxScene(DIM) for _ in range(int(os.environ.get("EXTRA_SCENES", "0")))]
for s_ in others:
    t_ = torch.empty(64 * 64 * 4, dtype=torch.uint8, device="cuda")
    f_ = ntracer_amd.ImageFormat(64, 64, [ntracer_amd.Channel(*c) for c in bench.RGBX8])._as_struct()
    _lib.check(L.nt_render_frames_device(s_._handle, C.c_void_p(t_.data_ptr()), 64 * 64 * 4, 1, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p),
                                         C.byref(f_), None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
NS = int(os.environ.get("NSTREAMS", "2"))            # streams of the "two streams" legs
scenes = [tracern.BoxScene(DIM) for _ in range(NS)]
fbs = [torch.empty((F, own * fmt.pitch), dtype=torch.uint8, device="cuda") for _ in range(NS)]
for _ in range(int(os.environ.get("SKIP_STREAMS", "0"))):          # (torch hands out pool streams in turn: start further down the pool)
    torch.cuda.Stream()
streams = [torch.cuda.Stream() for _ in range(NS)]
if os.environ.get("USE_NULL_STREAM", "0") != "0":
    streams[0] = torch.cuda.current_stream()
tabs = [L.nt_camera_table_create(DIM, F, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), 0) for _ in range(NS)]


def call(i, table, opts=opts):
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
                call(k % NS if two else 0, table, opts_ov if two else opts)
                k += 1
                if k % 8 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            steps = 200
            t0 = time.perf_counter()
            for k in range(steps):
                call(k % NS if two else 0, table, opts_ov if two else opts)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e6)
        print("world %d, %s, %s: %.1f us a step (best of 3: %s)" % (world, "camera table" if table else "host cameras", ("%d streams" % NS) if two else "one stream",
                                                                     min(res), ", ".join("%.1f" % r for r in res)))

#!/usr/bin/env python3
"""Can a render call be captured into a HIP graph (torch.cuda.graph) and replayed?  One rank's eighth of the 160 headline frames
from a camera table: one kernel launch a call, eight calls a graph.   python3 tools/graph_probe.py"""
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

g = np.load(os.path.join(ROOT, "tests", "golden", "box_n6_1920x1080.npz"))
F, W, H = 160, 1920, 1080
o = np.ascontiguousarray(g["origins"][:F], np.float32)
a = np.ascontiguousarray(g["axes"][:F], np.float32)
fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in bench.RGBX8])
fst = fmt._as_struct()
opts = _lib.NtRenderOpts()
opts.device = 0
opts.band_rank, opts.band_world, opts.band_rows, opts.compact = 0, 8, 8, 1
own = len(ntd.owned_rows(H, 0, 8, 8))
L = _lib.lib()
sc = tracern.BoxScene(6)
tab = L.nt_camera_table_create(6, F, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), 0)
fb = torch.zeros((F, own * fmt.pitch), dtype=torch.uint8, device="cuda")
ref = torch.zeros_like(fb)
s = torch.cuda.Stream()


def call(buf, stream):
    _lib.check(L.nt_render_table_device(sc._handle, C.c_void_p(buf.data_ptr()), own * fmt.pitch, C.c_void_p(tab), 0, F, C.byref(fst), C.byref(opts),
                                        C.c_void_p(stream.cuda_stream)))


with torch.cuda.stream(s):
    call(ref, s)                      # warm-up on the capture stream: scratch allocated, row table cached, no stream change later
    call(fb, s)
s.synchronize()
fb.zero_()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
K = 8
with torch.cuda.graph(gr, stream=s):
    for _ in range(K):
        call(fb, s)
gr.replay()
torch.cuda.synchronize()
print("graph of %d calls replayed: same bytes as the direct call: %s" % (K, bool(torch.equal(fb, ref))))
for name, fn, n in (("direct calls", lambda: [call(fb, s) for _ in range(K)], 40), ("graph replays", gr.replay, 40)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print("%s: %.1f us a call" % (name, (time.perf_counter() - t0) / (n * K) * 1e6))

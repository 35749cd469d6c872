#!/usr/bin/env python3
"""Per-wave residency trace of the packet kernel (needs a library built with -DNT_EXP_TRACE, passed through
NTRACER_HIP_LIB): writes gpurun_out/wave_trace.npz with (t0, t1, hw_id, xcc_id, hit mask) per tile wave."""
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
W, H = 1920, 1080
fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(8, 1, 0, 0), ntracer_amd.Channel(8, 0, 1, 0),
                                     ntracer_amd.Channel(8, 0, 0, 1), ntracer_amd.Channel(8, 0, 0, 0)])
fst = fmt._as_struct()
sel = [(i * 160) // frames for i in range(frames)]
o = np.ascontiguousarray(g["origins"][sel], np.float32)
a = np.ascontiguousarray(g["axes"][sel], np.float32)
tiles = ((W + 7) // 8) * ((H + 7) // 8)
extra = (tiles * frames * 32 + fmt.pitch * H - 1) // (fmt.pitch * H)
fb = torch.zeros((frames + extra, fmt.pitch * H), dtype=torch.uint8, device="cuda")
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
    print("%s: %.3f ms/frame" % (name, e0.elapsed_time(e1) / frames))
tr = fb[frames:].reshape(-1)[:tiles * frames * 32].cpu().numpy().view(np.uint64).reshape(frames, tiles, 4)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "wave_trace.npz"), trace=tr)
t0 = tr[..., 0].astype(np.int64); t1 = tr[..., 1].astype(np.int64)
base = t0.min()
print("span (100 MHz ticks)", t1.max() - base, "sum residency", (t1 - t0).sum(), "mean concurrency", (t1 - t0).sum() / (t1.max() - base))

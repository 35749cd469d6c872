#!/usr/bin/env python3
"""What becomes of the stretches of the bench frames (diagnostic; one GPU; needs a -DNT_DEBUG_SCRATCH build:
NTRACER_HIPCC_FLAGS=-DNT_DEBUG_SCRATCH python -m ntracer_amd.build /tmp/dbg.so; NTRACER_HIP_LIB=/tmp/dbg.so python3 tools/box_census.py).
Uses the cull / box / redo path (NTRACER_BOX_PATH=0), whose codes and redo words stay readable after the launch."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NTRACER_BOX_PATH"] = "0"
import torch
import ntracer_amd
from ntracer_amd import _lib, tracern

frames = 160
g = np.load(os.path.join(ROOT, "tests", "golden", "box_n6_1920x1080.npz"))
W, H, n = 1920, 1080, 6
origins = np.ascontiguousarray(g["origins"][:frames], np.float32)
axes = np.ascontiguousarray(g["axes"][:frames], np.float32)
fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(8, 1, 0, 0), ntracer_amd.Channel(8, 0, 1, 0), ntracer_amd.Channel(8, 0, 0, 1), ntracer_amd.Channel(8, 0, 0, 0)])
fst = fmt._as_struct()
L = _lib.lib()
L.nt_debug_box_scratch.restype = C.c_longlong
L.nt_debug_box_scratch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
scene = tracern.BoxScene(n)
fb = torch.empty((frames, H * fmt.pitch), dtype=torch.uint8, device="cuda")
_lib.check(L.nt_render_frames_device(scene._handle, C.c_void_p(fb.data_ptr()), H * fmt.pitch, frames, origins.ctypes.data_as(_lib.f32p),
                                     axes.ctypes.data_as(_lib.f32p), C.byref(fst), None, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
buf = np.zeros(64 << 20, np.uint8)
assert L.nt_debug_box_scratch(scene._handle, 0, buf.ctypes.data, buf.nbytes) > 0, _lib.last_error()
cols = (W + 63) // 64
words = (cols + 31) // 32
codes = buf[:frames * H * 4 * words * 4].view(np.uint32).reshape(frames, H, 4 * words)
redo = buf[(frames * H + 16) * 4 * words * 4:][:frames * H * words * 4].view(np.uint32).reshape(frames, H, words)
m = np.zeros((frames, H, cols), bool)
c = np.zeros((frames, H, cols), np.uint8)
for k in range(cols):
    m[:, :, k] = (redo[:, :, k // 32] >> (k % 32)) & 1
    c[:, :, k] = (codes[:, :, k // 8] >> (4 * (k % 8))) & 15
tot = c.size
print("stretches %d: culled %.2f %%, one face %.2f %%, code 14 %.2f %%, code 15 %.2f %%" % (tot, 100 * (c == 0).mean(), 100 * ((c >= 1) & (c <= 13)).mean(),
                                                                                    100 * (c == 14).mean(), 100 * (c == 15).mean()))
print("redo %.2f %% of all: from code 14 %.2f %%, from code 15 %.2f %%, from codes <= 13 (guard failures deferred) %.4f %%" % (
    100 * m.mean(), 100 * (m & (c == 14)).mean(), 100 * (m & (c == 15)).mean(), 100 * (m & (c <= 13)).mean()))
print("code 15 stretches that the box kernel settles itself: %.1f %%" % (100 * (~m & (c == 15)).sum() / max(1, (c == 15).sum())))
per_frame = m.reshape(frames, -1).mean(axis=1)
print("redo share per frame: min %.2f %% median %.2f %% max %.2f %%; frames above 10 %%: %d" % (100 * per_frame.min(), 100 * np.median(per_frame), 100 * per_frame.max(), int((per_frame > 0.10).sum())))

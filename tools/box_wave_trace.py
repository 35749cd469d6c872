#!/usr/bin/env python3
"""Per-wave residency trace of box_tile_kernel (needs a library built with -DNT_EXP_TRACE, passed through NTRACER_HIP_LIB):
the bench workload, F frames in one call; prints the occupancy timeline, the wave durations by class and what the end of the
kernel looks like, and writes gpurun_out/box_wave_trace.npz.   python3 tools/box_wave_trace.py [frames]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntracer_amd  # noqa: E402
from ntracer_amd import _lib, tracern  # noqa: E402
import bench  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 160
g = np.load(os.path.join(ROOT, "tests", "golden", "box_n6_1920x1080.npz"))
idx = np.arange(F) % len(g["origins"])
o = np.ascontiguousarray(g["origins"][idx], np.float32)
a = np.ascontiguousarray(g["axes"][idx], np.float32)
W, H = 1920, 1080
fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in bench.RGBX8])
fst = fmt._as_struct()
cols, trows = (W + 63) // 64, (H + 63) // 64
nrec = F * trows * cols
frame_bytes = fmt.pitch * H
extra = (nrec * 32 + frame_bytes - 1) // frame_bytes
fb = torch.zeros((F + extra, frame_bytes), dtype=torch.uint8, device="cuda")
sc = tracern.BoxScene(6)
st = torch.cuda.current_stream()
for rep in range(3):
    _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), frame_bytes, F, o.ctypes.data_as(_lib.f32p),
                                                  a.ctypes.data_as(_lib.f32p), C.byref(fst), None, C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
rec = fb[F:].reshape(-1)[: nrec * 32].cpu().numpy().view(np.uint64).reshape(F, trows, cols, 4)
t0, t1, hw, info = (rec[..., k].astype(np.int64) for k in range(4))
if not t1.any():
    sys.exit("no trace records: is NTRACER_HIP_LIB a -DNT_EXP_TRACE build?")
TICK = 0.01                      # s_memrealtime: 100 MHz
start = t0.min()
b = (t0 - start) * TICK          # us
e = (t1 - start) * TICK
dur = e - b
codes = ((info >> 32) & 0xffffffff) * TICK
rows = info & 0xffffffff
culled, face, rays, tie = rows & 255, (rows >> 8) & 255, (rows >> 16) & 255, (rows >> 24) & 255
span = e.max()
print("kernel span %.1f us, %d waves, wave duration mean %.1f us, median %.1f, p99 %.1f, max %.1f; codes phase mean %.2f us" %
      (span, dur.size, dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max(), codes.mean()))
heavy = rays + tie
for lo, hi in ((0, 0), (1, 8), (9, 24), (25, 48), (49, 64)):
    m = (heavy >= lo) & (heavy <= hi)
    if m.any():
        print("  waves with %2d..%2d per-ray rows: %6d (%.1f %%), duration mean %7.1f us, sum %.0f wave-us (%.1f %% of all)" %
              (lo, hi, m.sum(), 100.0 * m.mean(), dur[m].mean(), dur[m].sum(), 100.0 * dur[m].sum() / dur.sum()))
# occupancy timeline: waves in flight per 5 us
edges = np.arange(0.0, span + 5.0, 5.0)
occ = np.zeros(len(edges) - 1)
for k in range(len(occ)):
    lo, hi = edges[k], edges[k + 1]
    occ[k] = (np.clip(np.minimum(e, hi) - np.maximum(b, lo), 0, None)).sum() / (hi - lo)
print("waves in flight (average per 5 us slice; 8192 slots at 8 a SIMD, 7168 at 7):")
for k in range(0, len(occ), max(1, len(occ) // 40)):
    print("  t = %6.1f us: %6.0f  |%s" % (edges[k], occ[k], "#" * int(occ[k] / 160)))
last = np.sort(e.ravel())[::-1]
print("the last waves end at: " + ", ".join("%.1f" % v for v in last[:5]) + " us; 99 %% of the waves have ended by %.1f us, 99.9 %% by %.1f us" %
      (np.percentile(e, 99), np.percentile(e, 99.9)))
late = e > np.percentile(e, 99.5)
print("the last 0.5 %% of the waves: frames %d..%d, per-ray rows mean %.1f, duration mean %.1f us, started at %.1f us on average" %
      (np.nonzero(late)[0].min(), np.nonzero(late)[0].max(), heavy[late].mean(), dur[late].mean(), b[late].mean()))
fstart = b.reshape(F, -1).min(axis=1)
print("first wave of frame 0 / %d / %d starts at %.1f / %.1f / %.1f us" % (F // 2, F - 1, fstart[0], fstart[F // 2], fstart[F - 1]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "box_wave_trace.npz"), begin_us=b.astype(np.float32), end_us=e.astype(np.float32), hw=hw, rows=rows)

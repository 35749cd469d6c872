#!/usr/bin/env python3
"""BoxScene(n) 1920x1080 RGBX8, `frames` random cameras per call (seeded): ms per frame.  python3 tools/boxn_time.py n [frames [reps]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntracer_amd  # noqa: E402
from ntracer_amd import _lib, tracern  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1])
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(42 + n)
origins, axes = [], []
for k in range(frames):
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    q = np.ascontiguousarray(q, np.float32)
    origins.append((-q[2] * np.float32(4.0 * np.sqrt(n))).astype(np.float32))       # the scripts' distance: 4 sqrt(n)
    axes.append(q)
o = np.ascontiguousarray(np.stack(origins), np.float32)
a = np.ascontiguousarray(np.stack(axes), np.float32)
fmt = ntracer_amd.ImageFormat(1920, 1080, [ntracer_amd.Channel(*c) for c in bench.RGBX8])
ms = bench._time_frames(torch, _lib, tracern.BoxScene(n), fmt, o, a, frames, reps) / frames
print("BoxScene(%d) 1920x1080, %d frames a call: %.4f ms/frame = %.1f Grays/s%s" % (n, frames, ms, 1920 * 1080 / ms / 1e6,
      " (NTRACER_FORCE_VAR)" if os.environ.get("NTRACER_FORCE_VAR") else ""))

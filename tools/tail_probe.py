#!/usr/bin/env python3
"""Is the fixed part of a BoxScene call's time the drain of its long waves?  The bench rotation (cube in view: waves of very
different lengths) against the same cameras looking the other way (every tile culled: all waves alike), at several
frames per call.  python3 tools/tail_probe.py [frames ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntracer_amd  # noqa: E402
from ntracer_amd import _lib, tracern  # noqa: E402
import bench  # noqa: E402

counts = [int(a) for a in sys.argv[1:]] or [80, 160, 320]
g = np.load(os.path.join(ROOT, "tests", "golden", "box_n6_1920x1080.npz"))
origins, axes = g["origins"], g["axes"]
fmt = ntracer_amd.ImageFormat(1920, 1080, [ntracer_amd.Channel(*c) for c in bench.RGBX8])
for label, flip in (("cube in view", False), ("looking away", True)):
    for f in counts:
        idx = np.arange(f) % len(origins)
        o = np.ascontiguousarray(origins[idx], np.float32)
        a = np.ascontiguousarray(axes[idx], np.float32).copy()
        if flip:
            a[:, 2, :] = -a[:, 2, :]
        ms = bench._time_frames(torch, _lib, tracern.BoxScene(6), fmt, o, a, f, 10)
        print("%-13s %4d frames a call: %8.1f us  (%.3f us/frame)" % (label, f, ms * 1e3, ms * 1e3 / f), flush=True)

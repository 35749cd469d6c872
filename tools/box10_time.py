#!/usr/bin/env python3
"""BoxScene(n) at WxH, `frames` frames per call, from the golden cameras of box_n10_4096x4096.npz (any n <= 10 uses the first n
coordinates ... no: the file's own n): ms per frame.  python3 tools/box10_time.py [frames [reps]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntracer_amd  # noqa: E402
from ntracer_amd import _lib, tracern  # noqa: E402
import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = np.load(os.path.join(ROOT, "tests", "golden", "box_n10_4096x4096.npz"))
fmt = ntracer_amd.ImageFormat(4096, 4096, [ntracer_amd.Channel(*c) for c in bench.RGBX8])
ms = bench._time_frames(torch, _lib, tracern.BoxScene(10), fmt, g["origins"], g["axes"], frames, reps) / frames
print("BoxScene(10) 4096x4096, %d frames a call: %.3f ms/frame = %.1f Grays/s (NTRACER_BOX_R64=%s)" % (frames, ms, 4096 * 4096 / ms / 1e6, os.environ.get("NTRACER_BOX_R64", "-")))

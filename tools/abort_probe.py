#!/usr/bin/env python3
"""How long does nt_render take to come back after signal_abort()?  The 120-cell walked strictly at 4096x4096 (the abort test's
scene), the flag raised at several points of the frame.   python3 tools/abort_probe.py"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntracer_amd  # noqa: E402
from ntracer_amd import tracern  # noqa: E402
import fixtures as fx  # noqa: E402

g = fx.load("cell120_n4")
sc = tracern.CompositeScene.from_flat(4, fx.flat_of(g))
sc._set_camera_arrays(g["origins"][0], g["axes"][0])
fmt = ntracer_amd.ImageFormat(4096, 4096, [ntracer_amd.Channel(*c) for c in fx.RGBX8])
r = ntracer_amd.BlockingRenderer()
buf = bytearray(fmt.pitch * 4096)
for strict in (True, False):
    r.render(buf, fmt, sc, strict_reference=strict)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        r.render(buf, fmt, sc, strict_reference=strict)
        t.append(time.perf_counter() - t0)
    t_full = min(t)
    print("strict=%s: full frame %.2f ms" % (strict, t_full * 1e3))
    for frac in (0.05, 0.2, 0.4, 0.6):
        out = {}

        def run():
            out["ok"] = r.render(buf, fmt, sc, strict_reference=strict)
            out["end"] = time.perf_counter()
        th = threading.Thread(target=run)
        t_start = time.perf_counter()
        th.start()
        while not sc.locked and th.is_alive():
            pass
        time.sleep(frac * t_full)
        t_sig = time.perf_counter()
        r.signal_abort()
        th.join()
        print("   flag raised %.2f ms after the start: back after another %.2f ms (render returned %s)" %
              ((t_sig - t_start) * 1e3, (out["end"] - t_sig) * 1e3, out["ok"]))

import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntracer_amd
from ntracer_amd import tracern
import oracle_binding as ob
import fixtures as fx
g = fx.load("cell600_n4")
flat = fx.flat_of(g)
sc = tracern.CompositeScene.from_flat(4, flat)
sc._set_camera_arrays(g["origins"][5], g["axes"][5])
osc = ob.OracleScene(4, g["origins"][5], g["axes"][5], flat=flat)
for (w, h) in [(64, 64), (67, 45), (320, 200)]:
    fmt = ntracer_amd.ImageFormat(w, h, [ntracer_amd.Channel(*c) for c in fx.RGBX8])
    buf = bytearray(fmt.pitch * h)
    ntracer_amd.BlockingRenderer().render(buf, fmt, sc)
    got = np.frombuffer(bytes(buf), np.uint8).reshape(h, fmt.pitch)
    ref = osc.render(w, h, fx.RGBX8, threads=7)
    d = np.abs(got.astype(int) - ref.astype(int))
    print((w, h), "max byte delta", d.max(), "differing", int((d > 0).sum()))

import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ntracer_amd import tracern
import oracle_binding as ob
G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "feature3d.npz"))
flat = {k: g[k] for k in tracern._FLAT_KEYS}; flat["batch_size"] = 4
mats = g["materials"].copy(); mats[:, 6] = 1.0; mats[:, 7] = 0.0; flat["materials"] = mats
w, h = 96, 64
ys, xs = np.mgrid[0:h, 0:w]; xs = xs.ravel(); ys = ys.ravel()
sc = tracern.CompositeScene.from_flat(3, flat)
sc.set_fov(2.0)
for origin in ([0.3, 0.8, -7.0], [0.1, 0.5, 0.2], [0.0, -1.9999, 0.0], [0.0, -2.0, 0.0], [0.5, -2.0000002, 0.3], [2.0, 1.0, -2.0]):
    axes = np.eye(3, dtype=np.float32)
    o = np.asarray(origin, np.float32)
    sc._set_camera_arrays(o, axes)
    c = sc.colors_at(xs, ys, w, h)
    osc = ob.OracleScene(3, o, axes, fov=sc.fov, flat=flat)
    oc, cnt = osc.colors_at(xs, ys, w, h, counters=True)
    d = np.abs(c - oc).max(axis=1)
    print(origin, "mismatch", int((d > 1e-4).sum()), "max", d.max(), "oracle hits", cnt["hits"], "enter", cnt["aabb_enter"])

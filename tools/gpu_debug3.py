import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ntracer_amd import tracern
import oracle_binding as ob
G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "feature3d.npz"))
flat = {k: g[k] for k in tracern._FLAT_KEYS}; flat["batch_size"] = 4
mats = g["materials"].copy(); mats[:, 6] = 1.0; flat["materials"] = mats
flat_nr = dict(flat); m2 = mats.copy(); m2[:, 7] = 0; flat_nr["materials"] = m2
w, h = 96, 64
osc = ob.OracleScene(3, g["origin"], g["axes"], flat=flat)
sc = tracern.CompositeScene.from_flat(3, flat_nr)     # no reflection: plain trace
sc2 = tracern.CompositeScene.from_flat(3, flat)
for (x, y) in [(52, 43), (50, 43), (78, 43), (55, 58)]:
    dr = osc.primary_dir(x, y, w, h)
    r = osc.kd_intersects(g["origin"], dr, 0.0)
    nrm = r["normal"]; P = r["origin"]
    sine = -np.float32(np.dot(dr, nrm)); rd = (dr - nrm * (np.float32(-2) * sine)).astype(np.float32)
    # camera looking along rd from P
    a = np.array([1, 0, 0], np.float32) if abs(rd[0]) < 0.9 else np.array([0, 1, 0], np.float32)
    right = np.cross(rd, a); right /= np.linalg.norm(right); up = np.cross(right, rd)
    axes = np.stack([right, up, rd]).astype(np.float32)
    sc._set_camera_arrays(P, axes)
    c = sc.colors_at([48], [32], w, h)
    o2 = ob.OracleScene(3, P, axes, flat=flat_nr)
    oc = o2.colors_at([48], [32], w, h)
    d2 = o2.primary_dir(48, 32, w, h)
    hit = o2.kd_intersects(P, d2, 0.0)
    print((x, y), "P", P, "rd", rd, "d2", d2)
    print("   as primary: gpu", c, "oracle", oc, "oracle hit", None if hit is None else (hit["kind"], hit["index"], hit["lane"], hit["dist"]))
    # and the original pixel with reflection on the GPU
    sc2._set_camera_arrays(g["origin"], g["axes"])
    print("   with reflection: gpu", sc2.colors_at([x], [y], w, h), "oracle", osc.colors_at([x], [y], w, h))

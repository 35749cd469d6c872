import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntracer_amd
from ntracer_amd import tracern
import oracle_binding as ob
G = os.path.join(ROOT, "tests", "golden")
PK = ["fov", "shadows", "camera_light", "max_reflect_depth", "bg_gradient_axis", "ambient", "bg1", "bg2", "bg3",
      "point_light_pos", "point_light_color", "global_light_dir", "global_light_color"]
g = np.load(os.path.join(G, "feature3d.npz"))
flat = {k: g[k] for k in tracern._FLAT_KEYS}; flat["batch_size"] = 4
mats = g["materials"].copy(); mats[:, 6] = 1.0; flat["materials"] = mats
w, h = int(g["width"]), int(g["height"])
ys, xs = np.mgrid[0:h, 0:w]; xs = xs.ravel(); ys = ys.ravel()
sc = tracern.CompositeScene.from_flat(3, flat)
sc._set_camera_arrays(g["origin"], g["axes"])
params = {k: g["default__%s" % k] for k in PK}
for depth in (0, 1, 2, 4):
    params["max_reflect_depth"] = np.int32(depth)
    sc.set_params_flat(params)
    c = sc.colors_at(xs, ys, w, h)
    o = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=params).colors_at(xs, ys, w, h)
    d = np.abs(c - o).max(axis=1).reshape(h, w)
    print("depth", depth, "mismatch", int((d > 1e-4).sum()))
    if depth in (1, 4):
        for y in range(h):
            print("".join("#" if v > 1e-4 else "." for v in d[y]))
# which primitive do mismatching pixels hit on reflection? use the oracle per-stage API
params["max_reflect_depth"] = np.int32(1)
sc.set_params_flat(params)
c = sc.colors_at(xs, ys, w, h)
osc = ob.OracleScene(3, g["origin"], g["axes"], flat=flat, params=params)
o = osc.colors_at(xs, ys, w, h)
d = np.abs(c - o).max(axis=1)
for i in np.nonzero(d > 1e-4)[0][:12]:
    x, y = int(xs[i]), int(ys[i])
    dr = osc.primary_dir(x, y, w, h)
    r = osc.kd_intersects(g["origin"], dr, 0.0)
    nrm = r["normal"]; P = r["origin"]
    sine = -np.float32(np.dot(dr, nrm)); rd = (dr - nrm * (np.float32(-2) * sine)).astype(np.float32)
    r2 = osc.kd_intersects(P, rd, 0.0, skip_item=(r["index"] << 2) | r["kind"], skip_lane=r["lane"])
    print((x, y), "primary", r["kind"], r["index"], r["lane"], "refl hit", None if r2 is None else (r2["kind"], r2["index"], r2["lane"], r2["dist"]), "gpu", c[i], "orc", o[i])

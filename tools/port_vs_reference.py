#!/usr/bin/env python3
"""How fast is the CPU port (oracle/) next to the reference itself?  BUILD CONTAINER ONLY: the reference (built in a scratch
directory outside the repo with its documented setup.py options) and the port render the same frames on the
same cores, alternately, with the reference's thread rule (BlockingRenderer(-1): hardware_concurrency() - 1 workers + the
caller).  Writes profiles/port_vs_reference.json, the constant bench.py carries in `cpu_baseline.port_vs_reference`.

    PYTHONPATH=/tmp/ntracer_oracle/build/lib.linux-x86_64-3.10 python3 tools/port_vs_reference.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_golden as gg  # noqa: E402  (imports the reference; provides set_cam / polytope_scene)
from ntracer import NTracer, ImageFormat, BlockingRenderer  # noqa: E402
import oracle_binding as ob  # noqa: E402

W, H = 1920, 1080
RGBX8 = [(8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1), (8, 0, 0, 0)]
G = os.path.join(ROOT, "tests", "golden")


def time_reference(nt, scene, origins, axes, frames, reps):
    fmt = ImageFormat(W, H, gg.RGBX8())
    buf = bytearray(fmt.pitch * H)
    r = BlockingRenderer()                    # threads = -1: the default
    best = []
    for _ in range(reps):
        t = []
        for f in frames:
            gg.set_cam(nt, scene, origins[f], axes[f])
            t0 = time.perf_counter()
            assert r.render(buf, fmt, scene)
            t.append(time.perf_counter() - t0)
        best.append(t)
    return np.array(best)


def time_port(osc, origins, axes, frames, reps):
    r = ob.OracleRenderer(-1)
    best = []
    for _ in range(reps):
        t = []
        for f in frames:
            osc.set_camera(origins[f], axes[f])
            t0 = time.perf_counter()
            r.render(osc, W, H, RGBX8)
            t.append(time.perf_counter() - t0)
        best.append(t)
    n = r.threads
    r.close()
    return np.array(best), n


def main():
    out = {"where": "build container (%d CPUs, %s)" % (os.cpu_count(), open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")),
           "reference_build": "python3 setup.py build --cpp-neg-opts=-march=native --cpp-opts='-march=nehalem -include stddef.h' (SSE4.2: this commit's "
                              "AVX paths do not compile, SURVEY.md 8c), g++ -O3 -ffast-math; BATCH_SIZE 4",
           "port_build": "oracle/Makefile: gcc -O2 -ffp-contract=off, scalar C",
           "command": "PYTHONPATH=<reference build> python3 tools/port_vs_reference.py",
           "method": "same 1920x1080 RGBX8 frames, reference BlockingRenderer() and the port's nto_renderer alternately on the same cores, "
                     "threads = hardware_concurrency() (workers + caller) for both; per config the best time of each frame over the repetitions, summed"}
    # ---- BoxScene(6), configs[2]
    g = np.load(os.path.join(G, "box_n6_1920x1080.npz"))
    nt = NTracer(6)
    scene = nt.BoxScene()
    frames = [0, 20, 40, 60, 80, 100, 120, 140]
    tr, tp = [], []
    osc = ob.OracleScene(6, g["origins"][0], g["axes"][0])
    for rep in range(3):
        tr.append(time_reference(nt, scene, g["origins"], g["axes"], frames, 1)[0])
        a, threads = time_port(osc, g["origins"], g["axes"], frames, 1)
        tp.append(a[0])
    tr, tp = np.min(tr, axis=0), np.min(tp, axis=0)
    out["box6_1080p"] = {"reference_Mrays_s": round(W * H * len(frames) / tr.sum() / 1e6, 2), "port_Mrays_s": round(W * H * len(frames) / tp.sum() / 1e6, 2),
                         "port_over_reference": round(float(tr.sum() / tp.sum()), 3), "threads": threads, "frames": frames}
    print(out["box6_1080p"], flush=True)
    # ---- the 120-cell, configs[3]: the reference renders the scene its own script builds, the port the captured fixture of
    # an earlier run of the same script (the script's output order is hash-order dependent; the work per ray is the same)
    nt4, scene4, cam_distance = gg.polytope_scene(["5/2", "3", "3"])
    g4 = np.load(os.path.join(G, "cell120_n4.npz"))
    flat = {k: g4[k] for k in ("root", "node_axis", "node_split", "node_left", "node_right", "items", "batch_recs", "batch_mats", "tri_recs", "tri_mats",
                               "solid_recs", "solid_types", "solid_mats", "materials", "aabb_start", "aabb_end")}
    flat["batch_size"] = 4
    osc4 = ob.OracleScene(4, g4["origins"][0], g4["axes"][0], flat=flat)
    frames = [0, 40]
    tr, tp = [], []
    for rep in range(2):
        tr.append(time_reference(nt4, scene4, g4["origins"], g4["axes"], frames, 1)[0])
        a, threads = time_port(osc4, g4["origins"], g4["axes"], frames, 1)
        tp.append(a[0])
    tr, tp = np.min(tr, axis=0), np.min(tp, axis=0)
    out["cell120_1080p"] = {"reference_Mrays_s": round(W * H * len(frames) / tr.sum() / 1e6, 3), "port_Mrays_s": round(W * H * len(frames) / tp.sum() / 1e6, 3),
                            "port_over_reference": round(float(tr.sum() / tp.sum()), 3), "threads": threads, "frames": frames}
    print(out["cell120_1080p"], flush=True)
    out["reading"] = "port_over_reference < 1: the port is slower than the reference, so a GPU/CPU ratio read off bench.py's cpu_baseline (kind \"port\") " \
                     "overstates the ratio to the reference's own CPU path by 1 / port_over_reference"
    json.dump(out, open(os.path.join(ROOT, "profiles", "port_vs_reference.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

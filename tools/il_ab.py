#!/usr/bin/env python3
"""A/B of environment switches the library reads at every call, interleaved rounds in ONE process (cdna_hip_programming.md
5.4 rule 24), on the BoxScene launches the bench times.  Also checks that both settings render the same bytes.

    python3 tools/il_ab.py [--var NTRACER_BOX_INTERLEAVE] [--values 0,1] [--rounds 6] [--steps 10] [--cases head,f32,band8,box3,box10]

One JSON line per case: median / min microseconds per call (HIP events on the launch stream) for each value."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RGBX8 = [(8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1), (8, 0, 0, 0)]
RGBF32 = [(32, 1, 0, 0, 0, True), (32, 0, 1, 0, 0, True), (32, 0, 0, 1, 0, True)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--var", default="NTRACER_BOX_INTERLEAVE")
    ap.add_argument("--values", default="0,1")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cases", default="head,f32,band8,box3,box10")
    ap.add_argument("--frames", type=int, default=160)
    a = ap.parse_args()
    import torch
    import ntracer_amd
    from ntracer_amd import _lib, tracern
    from ntracer_amd import distributed as ntd
    L = _lib.lib()
    st = torch.cuda.current_stream()
    values = a.values.split(";") if ";" in a.values else a.values.split(",")

    def case(name):
        world, rank, brows = 1, 0, 32
        chans, n, W, H, F = RGBX8, 6, 1920, 1080, a.frames
        if name == "f32":
            chans = RGBF32
        elif name == "band8":
            world, brows = 8, 8
        elif name == "box3":
            n = 3
        elif name == "box10":
            n, W, H, F = 10, 4096, 4096, 16
        elif name == "head80":
            F = 80
        elif name == "head320":
            F = 320
        g = np.load(os.path.join(ROOT, "tests", "golden", "box_n%d_%dx%d.npz" % (n, W, H)))
        idx = np.arange(F) % len(g["origins"])
        o = np.ascontiguousarray(g["origins"][idx], np.float32)
        ax = np.ascontiguousarray(g["axes"][idx], np.float32)
        fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in chans])
        fst = fmt._as_struct()
        opts = _lib.NtRenderOpts()
        opts.device = torch.cuda.current_device()
        opts.band_rank, opts.band_world, opts.band_rows, opts.compact = rank, world, brows, 1
        own = len(ntd.owned_rows(H, rank, world, brows))
        fb = torch.zeros((F, own * fmt.pitch), dtype=torch.uint8, device="cuda")
        sc = tracern.BoxScene(n)

        def go():
            _lib.check(L.nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), own * fmt.pitch, F, o.ctypes.data_as(_lib.f32p),
                                                 ax.ctypes.data_as(_lib.f32p), C.byref(fst), C.byref(opts), C.c_void_p(st.cuda_stream)))
        sums = {}
        for v in values:
            os.environ[a.var] = v
            fb.zero_()
            go()
            torch.cuda.synchronize()
            # a position-weighted checksum of every frame (int64 arithmetic on the device)
            x = fb.view(torch.int32).to(torch.int64)
            w = torch.arange(1, x.shape[1] + 1, device="cuda", dtype=torch.int64)
            sums[v] = [int(t) for t in ((x * w).sum(dim=1) & 0x7fffffffffff).cpu()]
        same = all(sums[v] == sums[values[0]] for v in values)
        times = {v: [] for v in values}
        import time
        t_end = time.perf_counter() + 0.3             # let the chip settle under the load first ("Settling")
        while time.perf_counter() < t_end:
            go()
            torch.cuda.synchronize()
        for _ in range(a.rounds):
            for v in values:
                os.environ[a.var] = v
                go()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(a.steps):
                    go()
                e1.record(st)
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3 / a.steps)
        out = {"case": name, "var": a.var, "same_bytes": same}
        for v in values:
            out["us_median[%s]" % v] = round(float(np.median(times[v])), 1)
            out["us_min[%s]" % v] = round(float(np.min(times[v])), 1)
        print(json.dumps(out), flush=True)
        del fb
        torch.cuda.empty_cache()

    for c in a.cases.split(","):
        case(c)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What one rank of an N-GPU run does, measured on one GPU: BoxScene(6) 1920x1080 RGBX8, 160 frames per call, rows dealt in
bands to `world` ranks, this process rendering rank `rank`'s bands only (compact buffer) -- exactly the per-rank workload
of `bench.py --gpus N`.

    python3 tools/band_proxy.py [--world 8] [--rank 0] [--band-rows 8] [--steps 60] [--warmup 10]

Prints one JSON line: HIP-event and host-clock microseconds per call."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(world, rank, band_rows, steps, warmup, frames=160, n=6, W=1920, H=1080, f32=False, overlapped=False):
    import torch
    import ntracer_amd
    from ntracer_amd import _lib, tracern
    from ntracer_amd import distributed as ntd
    g = np.load(os.path.join(ROOT, "tests", "golden", "box_n%d_1920x1080.npz" % n))
    idx = np.arange(frames) % len(g["origins"])               # (more frames than the rotation has: round again)
    origins = np.ascontiguousarray(g["origins"][idx], np.float32)
    axes = np.ascontiguousarray(g["axes"][idx], np.float32)
    scene = tracern.BoxScene(n)
    if f32:
        fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(32, 1, 0, 0, 0, True), ntracer_amd.Channel(32, 0, 1, 0, 0, True),
                                             ntracer_amd.Channel(32, 0, 0, 1, 0, True)])
    else:
        fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(8, 1, 0, 0), ntracer_amd.Channel(8, 0, 1, 0),
                                             ntracer_amd.Channel(8, 0, 0, 1), ntracer_amd.Channel(8, 0, 0, 0)])
    fst = fmt._as_struct()
    opts = _lib.NtRenderOpts()
    opts.device = torch.cuda.current_device()
    opts.band_rank = rank
    opts.band_world = world
    opts.band_rows = band_rows
    opts.compact = 1
    opts.overlapped = 1 if overlapped else 0        # (the launch shape of callers that overlap their calls; here one call at a time)
    own = len(ntd.owned_rows(H, rank, world, band_rows))
    fb = torch.empty((frames, own * fmt.pitch), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    L = _lib.lib()

    def go():
        _lib.check(L.nt_render_frames_device(scene._handle, C.c_void_p(fb.data_ptr()), own * fmt.pitch, frames,
                                             origins.ctypes.data_as(_lib.f32p), axes.ctypes.data_as(_lib.f32p), C.byref(fst), C.byref(opts),
                                             C.c_void_p(st.cuda_stream)))
    for _ in range(warmup):
        go()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(st)
    for _ in range(steps):
        go()
    e1.record(st)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return {"format": "rgbf32" if f32 else "rgbx8", "n": n, "box_path": os.environ.get("NTRACER_BOX_PATH", "1"), "world": world, "rank": rank, "band_rows": band_rows, "owned_rows": own, "frames": frames,
            "event_us_per_call": round(e0.elapsed_time(e1) * 1e3 / steps, 2), "wall_us_per_call": round(wall * 1e6 / steps, 2),
            "host_issue_us_per_call": round(t_issue * 1e6 / steps, 2)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=160)
    ap.add_argument("--n", type=int, default=6)
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--overlapped", action="store_true", help="nt_render_opts.overlapped = 1: the launch shape of the two-stream legs")
    a = ap.parse_args()
    print(json.dumps(measure(a.world, a.rank, a.band_rows, a.steps, a.warmup, frames=a.frames, n=a.n, f32=a.f32, overlapped=a.overlapped)))

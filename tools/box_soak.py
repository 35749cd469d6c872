#!/usr/bin/env python3
"""BoxScene soak (development aid): random cameras, full 1080p frames rendered in one multi-frame launch, every frame
compared with the oracle byte for byte.  python3 tools/box_soak.py [frames_per_dimension [rgbx8|rgbf32 [seed [dims]]]]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import ntracer_amd  # noqa: E402
from ntracer_amd import _lib, tracern  # noqa: E402
import oracle_binding as ob  # noqa: E402

RGBX = [(8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1), (8, 0, 0, 0)]


def main():
    per_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    chans = RGBX if len(sys.argv) < 3 or sys.argv[2] == "rgbx8" else [(32, 1, 0, 0, 0, True), (32, 0, 1, 0, 0, True), (32, 0, 0, 1, 0, True)]
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 900
    dims = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [3, 4, 6, 8, 10]
    w, h = 1920, 1080
    fmt = ntracer_amd.ImageFormat(w, h, [ntracer_amd.Channel(*c) for c in chans])
    st = fmt._as_struct()
    sys.path.insert(0, ROOT)
    import bench
    threads = max(1, min(200, bench.cpu_quota_cores() - 1))
    bad = 0
    RGBX_ = chans
    for n in dims:
        rng = np.random.default_rng(seed + n)
        origins, axes = [], []
        for k in range(per_dim):
            q, _ = np.linalg.qr(rng.standard_normal((n, n)))
            q = np.ascontiguousarray(q, np.float32)
            dist = float(rng.choice([1.2, 1.6, 2.5, 4.0, 7.0, 12.0]))
            o = -q[2] * np.float32(dist) + np.float32(rng.uniform(-0.6, 0.6)) * q[0] + np.float32(rng.uniform(-0.6, 0.6)) * q[1]
            if k % 7 == 0:                       # on a diagonal: coordinates equal up to rounding, like the demo path
                o = np.full(n, -dist / np.sqrt(n), np.float32)
                q = q.copy()
                q[2] = -o / np.linalg.norm(o)
            origins.append(o.astype(np.float32))
            axes.append(q)
        o = np.ascontiguousarray(np.stack(origins), np.float32)
        a = np.ascontiguousarray(np.stack(axes), np.float32)
        sc = tracern.BoxScene(n)
        # BANDS=world,rank,band_rows: one rank's share of the frames (compact buffer); OVERLAPPED=1: in the launch shape of callers
        # that overlap their calls (nt_render_opts.overlapped)
        own = np.arange(h)
        opts = None
        if os.environ.get("BANDS") or os.environ.get("OVERLAPPED"):
            from ntracer_amd import distributed as ntd
            world, rank, brows = (int(v) for v in os.environ.get("BANDS", "1,0,32").split(","))
            opts = _lib.NtRenderOpts()
            opts.device = -1
            opts.band_rank, opts.band_world, opts.band_rows, opts.compact = rank, world, brows, 1
            opts.overlapped = int(os.environ.get("OVERLAPPED", "0"))
            own = ntd.owned_rows(h, rank, world, brows)
        fb = torch.zeros((per_dim, len(own) * fmt.pitch), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().nt_render_frames_device(sc._handle, C.c_void_p(fb.data_ptr()), len(own) * fmt.pitch, per_dim,
                                                      o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(st), C.byref(opts) if opts is not None else None,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        got = fb.cpu().numpy().reshape(per_dim, len(own), fmt.pitch)
        osc = ob.OracleScene(n, o[0], a[0])
        nbad = 0
        for f in range(per_dim):
            osc.set_camera(o[f], a[f])
            ref = osc.render(w, h, RGBX_, threads=threads)
            d = int((got[f] != ref[own]).sum())
            if d:
                nbad += 1
                print("n=%d frame %d: %d bytes differ" % (n, f, d))
        print("n=%d: %d frames, %d with differences" % (n, per_dim, nbad), flush=True)
        bad += nbad
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

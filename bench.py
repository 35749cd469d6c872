#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ray-cast path on MI355X.

Metric (BASELINE.json): Mrays/s (primary+shadow), 6-D hypercube @1920x1080.
Workload (configs[2]): BoxScene(6) -- the scene of the reference's scripts/hypercube.py and of
`polytope.py 4 3 3 3 3` -- rendered at 1920x1080 into the RGBX8 format pygame surfaces use, over the
160-frame RotatingCamera sequence of scripts/polytope.py:522-556 (cameras captured from the reference,
tests/golden/box_n6_1920x1080.npz).  As scripted there are no lights and no shadows, so every ray is a
primary ray.

A step = one pass over the benchmark's batch of synthetic input: the 160 cameras of the rotation, i.e. 160 frames
= 331 776 000 primary rays, issued as ONE multi-frame launch (nt_render_frames_device: one camera and one
framebuffer per frame, all 1.3 GB resident in HBM).  K steps are timed between barrier +
torch.cuda.synchronize() pairs, with HIP events on the launch stream for the kernel time.

Settling.  An MI355X that has been idle drops its clock when a sustained load arrives and takes about 50 ms to settle
(tools/settle_probe.py: 0.43-0.47 ms a step during the first 10 ms, 0.355 ms from 50 ms on and for the next 8 s).  W warm-up
steps are 2 ms.  So after the W warm-up steps untimed steps go on back to back for --settle-ms (default 200) before EXACTLY K
steps are timed: `value` is the sustained rate.  What the first K steps after the W warm-up steps take -- how rounds 1 and 2
measured -- is reported beside it as `cold_start` (same steps, same work; the difference is the chip's clock).  Every other
timed region of this file (fp32 format, one rank's bands, config 5, the extras) is settled the same way.

N > 1 (one process per GPU, launched by torch.distributed.run): every frame is tiled across the ranks in
row bands (band b -> rank b % N; 32 rows = the reference's chunk size, or 16 / 8 rows when that shares the
rows out more evenly: 1080 rows over 8 ranks are 5-vs-4 bands of 32 rows but 17-vs-16 bands of 8).  Pixels are independent: no collective in the timed region; total
work is fixed, so scaling is "strong".  The RCCL gather of the finished bands to rank 0 is measured
separately after the timed region and reported as gather_ms_per_frame / value_incl_gather (like the
D2H copy at N = 1 it is a delivery step, not part of `value`).

One JSON line on rank 0, with the `roofline` of the dominant kernel (framebuffer write bytes vs HBM peak; `roofline.valu`:
the instruction-issue bound that actually limits these kernels, from the committed SQ counters), `value_rgbf32` (the same
workload into three fp32 channels -- the format in which "colours within 1e-4" is a statement about the stored values),
`scaling_proxy` (one rank's share of an 8-GPU run, timed on this GPU) and a `cpu_baseline` (the oracle, multi-threaded
exactly like the reference's BlockingRenderer, on a bounded sample of the same frames).

Config ids in `extra` are indices into BASELINE.json's `configs` list (0-based: cfg1 = 3-D hypercube 1080p, cfg2 = the
headline 6-D hypercube, cfg3 = 120-cell, cfg4 = 10-D hypercube 4096x4096).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)
STORE_ONLY_GBS = 5650.0       # 64 x 64-pixel tiles, 256 bytes per store instruction: tools/micro/store_rate.hip on MI355X
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
PROFILE = "r03_pmc_summary.json"       # profiles/: rocprofv3 --pmc passes of THIS build (tools/profile_round.sh)
RGBX8 = [(8, 1, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1), (8, 0, 0, 0)]
RGBF32 = [(32, 1, 0, 0, 0, True), (32, 0, 1, 0, 0, True), (32, 0, 0, 1, 0, True)]


def load_profile():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", PROFILE)))
    except (OSError, ValueError):
        return None


def valu_bound(call, measured_us):
    """The instruction-issue bound of a call from the committed rocprofv3 passes (profiles/<PROFILE>, tools/profile_round.sh):
    VALU wave-instructions by class (SQ_INSTS_VALU_ADD_F32 / MUL / FMA / TRANS / INT32 / INT64 / CVT, rest = total - these),
    priced with the issue costs tools/micro/valu_rate2.hip measures on this part -- 2 cycles per SIMD for full-rate arithmetic,
    4 for the half-rate instructions (compares, selects, min / max, cvt, fract, perm ...), 8 for transcendentals; INT32 and the
    rest are mixtures and are priced at 2 (`lo`) and at 4 (`hi`) -- against the kernel's own cycle count from the same profile
    (GRBM_GUI_ACTIVE / 8).  Cycles against cycles: no clock enters."""
    if not call:
        return None
    lo = call.get("issue_cycles_per_simd_lo")
    hi = call.get("issue_cycles_per_simd_hi")
    cyc = call.get("kernel_cycles_per_call", call.get("kernel_cycles"))
    valu = call.get("valu_wave_insts_per_call", call.get("SQ_INSTS_VALU"))
    if not lo or not hi or not cyc:
        return None
    salu = call.get("salu_wave_insts_per_call", call.get("SQ_INSTS_SALU"))
    return {"valu_wave_insts": round(valu), "salu_wave_insts": round(salu) if salu else None, "by_class": call.get("valu_by_class"),
            "price_cycles_per_wave_inst_per_simd": {"full rate (ADD/MUL/FMA f32)": 2, "TRANS f32": 8, "CVT": 4, "INT32, rest: lo": 2, "INT32, rest: hi": 4},
            "issue_limited_cycles_per_simd": [round(lo), round(hi)], "kernel_cycles_profiled": round(cyc),
            "frac_lo": round(lo / cyc, 4), "frac_hi": round(hi / cyc, 4), "frac": round(0.5 * (lo + hi) / cyc, 4),
            "issue_limited_us": [round(measured_us * lo / cyc, 1), round(measured_us * hi / cyc, 1)], "measured_us": round(measured_us, 2),
            "note": "frac = issue-limited cycles / kernel cycles, both from the profile (lo: every instruction of the mixed classes at the full "
                    "rate; hi: at half rate; frac: their mean); issue_limited_us scales the live launch time by the same ratios",
            "source": "profiles/" + PROFILE}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=160)
    ap.add_argument("--settle-ms", type=float, default=200.0, help="untimed steps are issued back to back for at least this long before every timed region: "
                    "the chip takes ~50 ms of sustained load to settle its clock (tools/settle_probe.py); 0: time from a cold start")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="only the timed headline loop (the target of the rocprofv3 --pmc passes)")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the N-rank plumbing (launcher, band split, barriers, gather over gloo): "
                    "no GPU, no kernels, no throughput -- used by tests/test_bench_launch.py")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        # a bare `python3 bench.py --gpus N`: start the N ranks ourselves -- fresh child processes, before this one has
        # loaded the library or touched the GPU -- and leave with their status; rank 0's JSON line goes straight to our stdout
        sys.exit(self_launch(args.gpus))
    if world != max(args.gpus, 1):
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.dry_run:
        return dry_run(args, world, rank)
    global SETTLE_MS
    SETTLE_MS = args.settle_ms

    import torch
    import ntracer_amd
    from ntracer_amd import _lib, tracern
    from ntracer_amd import distributed as ntd

    # rehearsal hooks (used only to exercise the N>1 code path on a 1-GPU box): several ranks on one device
    # cannot use RCCL, so NTRACER_BENCH_BACKEND=gloo NTRACER_BENCH_DEVICE=0 runs the same code over gloo
    backend = os.environ.get("NTRACER_BENCH_BACKEND", "nccl")
    if "NTRACER_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["NTRACER_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    g = np.load(os.path.join(ROOT, "tests", "golden", "box_n6_1920x1080.npz"))
    n, W, H = 6, 1920, 1080
    origins = np.ascontiguousarray(g["origins"], np.float32)        # [160][6]
    axes = np.ascontiguousarray(g["axes"], np.float32)              # [160][6][6]
    nrot = len(origins)
    scene = tracern.BoxScene(n)
    fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in RGBX8])
    fst = fmt._as_struct()
    opts = _lib.NtRenderOpts()
    opts.device = local_rank
    opts.band_rank = rank
    opts.band_world = world
    opts.compact = 1
    opts.strict_reference = 0
    band_rows = pick_band_rows(ntd, H, world)
    opts.band_rows = band_rows
    own_rows = len(ntd.owned_rows(H, rank, world, band_rows))
    frame_bytes = own_rows * fmt.pitch
    F = max(1, min(args.frames_per_step, nrot))
    fb = torch.empty((F, frame_bytes), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    L = _lib.lib()

    # The step's inputs -- the 160 cameras of the rotation -- are resident in device memory when the timed region starts, as the
    # bench contract asks (nt_camera_table_create packs and uploads them once; a call is then ONE kernel launch).  The same steps
    # with the cameras handed over as host arrays on every call, as the reference's boundary does (packed, uploaded and launched
    # each time: the PCIe-inclusive rate), are timed beside it and reported as `host_cameras`; they were `value` until round 3's
    # last day.  Everything a call needs is prepared once: the timed loop is the call itself, not numpy indexing around it.
    step_o = np.ascontiguousarray(origins[np.arange(F) % nrot])
    step_a = np.ascontiguousarray(axes[np.arange(F) % nrot])
    tab = L.nt_camera_table_create(n, F, step_o.ctypes.data_as(_lib.f32p), step_a.ctypes.data_as(_lib.f32p), local_rank)
    if not tab:
        sys.exit("nt_camera_table_create failed: " + _lib.last_error())
    call_args = (scene._handle, C.c_void_p(fb.data_ptr()), frame_bytes, C.c_void_p(tab), 0, F, C.byref(fst), C.byref(opts), C.c_void_p(stream.cuda_stream))
    host_args = (scene._handle, C.c_void_p(fb.data_ptr()), frame_bytes, F, step_o.ctypes.data_as(_lib.f32p), step_a.ctypes.data_as(_lib.f32p),
                 C.byref(fst), C.byref(opts), C.c_void_p(stream.cuda_stream))
    render_table = L.nt_render_table_device
    render_frames = L.nt_render_frames_device

    def run(steps):
        for _ in range(steps):
            r = render_table(*call_args)
            if r < 0:
                _lib.check(r)
        return steps

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_leg(step_fn, warm):
        """`warm` untimed steps, settling, then EXACTLY K steps between two barriers; MAX over ranks; ms per step."""
        for _ in range(warm):
            step_fn()
        settle(torch, step_fn, args.settle_ms)
        barrier()
        t_ = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        barrier()
        tt = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()) * 1e3 / args.steps

    # ---- cold start: the first `steps` steps after `warmup` steps from an idle chip (how rounds 1 and 2 measured).  The chip is
    # then in the middle of a clock transient -- it drops its clock when the load arrives and takes ~50 ms to settle
    # (tools/settle_probe.py: 0.43-0.47 ms a step in the first 10 ms, 0.355 from 50 ms on, flat for the next 8 s) -- so this is
    # reported as `cold_start`, and `value` is measured after `--settle-ms` of sustained load.
    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    cold = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(cold, op=dist.ReduceOp.MAX)
    cold_ms = float(cold.item()) * 1e3 / args.steps
    settle(torch, lambda: run(1), args.settle_ms)
    barrier()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    launches = run(args.steps)
    e1.record(stream)
    issue_s = time.perf_counter() - t0                 # the host's share: K calls issued (the GPU is still working)
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    el = torch.tensor([wall], dtype=torch.float64, device="cuda")
    dv = torch.tensor([dev_ms], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(dv, op=dist.ReduceOp.MAX)
    wall = float(el.item())
    dev_ms = float(dv.item())

    rays = float(W) * H * F * args.steps
    value = rays / wall / 1e6
    one_stream_value = value
    one_stream_ms = wall * 1e3 / args.steps

    # ---- the same K steps issued alternately on TWO streams (two scene handles -- each has its own device-side scratch -- and two
    # sets of framebuffers; the caller says so, nt_render_opts.overlapped, and the library shapes its launches for it: waves of 64
    # rows, whose long tail is what a call that runs alone cannot afford, from 64 rows up): the tail of one call overlaps the ramp
    # of the next.  For a full frame that is 2-3 % (the kernel is bound by vector issue); for a rank's share of a tiled frame it
    # is a quarter (tools/two_stream_probe.py: a rank's eighth 60.6 -> 46.0 us a step).  At N = 1 `value` stays the one-stream
    # figure (one kernel at a time: what the roofline and the committed profiles describe); at N > 1 the better of the two IS
    # `value` (`config.issue` says which).  EXACTLY K steps, barrier-bracketed, MAX over ranks, like the loop above.
    two_ms = host_ms = host_two_ms = None
    if not args.headline_only:
        scene2 = tracern.BoxScene(n)
        fb2 = torch.empty((F, frame_bytes), dtype=torch.uint8, device="cuda")
        stream2 = torch.cuda.Stream()
        opts2 = _lib.NtRenderOpts()
        C.memmove(C.byref(opts2), C.byref(opts), C.sizeof(opts))
        opts2.overlapped = 1
        tcalls = ((scene._handle, C.c_void_p(fb.data_ptr()), frame_bytes, C.c_void_p(tab), 0, F, C.byref(fst), C.byref(opts2), C.c_void_p(stream.cuda_stream)),
                  (scene2._handle, C.c_void_p(fb2.data_ptr()), frame_bytes, C.c_void_p(tab), 0, F, C.byref(fst), C.byref(opts2), C.c_void_p(stream2.cuda_stream)))
        hcalls = (host_args[:7] + (C.byref(opts2),) + host_args[8:],
                  (scene2._handle, C.c_void_p(fb2.data_ptr()), frame_bytes, F, step_o.ctypes.data_as(_lib.f32p), step_a.ctypes.data_as(_lib.f32p),
                   C.byref(fst), C.byref(opts2), C.c_void_p(stream2.cuda_stream)))
        state = {"k": 0}

        def step_table_two():
            r = render_table(*tcalls[state["k"] & 1])
            state["k"] += 1
            if r < 0:
                _lib.check(r)

        def step_host():
            r = render_frames(*host_args)
            if r < 0:
                _lib.check(r)

        def step_host_two():
            r = render_frames(*hcalls[state["k"] & 1])
            state["k"] += 1
            if r < 0:
                _lib.check(r)
        two_ms = timed_leg(step_table_two, 2 * args.warmup)
        # ---- the PCIe-inclusive way: the cameras as host arrays on every call (packed, uploaded, launched), one stream and two
        host_ms = timed_leg(step_host, args.warmup)
        host_two_ms = timed_leg(step_host_two, 2 * args.warmup)
        torch.cuda.synchronize()
        del fb2
    # at N > 1 both ways of issuing the steps were timed the same way and the better one IS `value` (`config.issue` says which);
    # at N = 1 `value` stays the one-stream figure the roofline and the committed profiles describe
    issue = "one stream"
    if world > 1 and two_ms is not None and two_ms * 1e-3 * args.steps < wall:
        wall = two_ms * 1e-3 * args.steps
        value = float(W) * H * F / (two_ms * 1e-3) / 1e6
        issue = "steps alternate between two streams (see two_streams)"

    # ---- delivery step, outside `value`: gather to rank 0 (RCCL) / D2H at N = 1
    gather_ms = None
    gather_ok = None
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
        reps = 5
        t1 = time.perf_counter()
        for i in range(reps):
            full = ntd.gather_framebuffer(fb[i % F], fmt, rank, world, dst=0, band_rows=band_rows)
        torch.cuda.synchronize()
        dist.barrier()
        gather_ms = (time.perf_counter() - t1) / reps * 1e3
        # the gathered frame must equal the same frame rendered whole by rank 0
        gather_ok = None
        if rank == 0:
            whole = torch.empty(H * fmt.pitch, dtype=torch.uint8, device="cuda")
            fidx = (reps - 1) % F                                         # camera of the slot gathered last
            o1 = np.ascontiguousarray(origins[[fidx % nrot]])
            a1 = np.ascontiguousarray(axes[[fidx % nrot]])
            _lib.check(L.nt_render_frames_device(scene._handle, C.c_void_p(whole.data_ptr()), H * fmt.pitch, 1,
                                                 o1.ctypes.data_as(_lib.f32p), a1.ctypes.data_as(_lib.f32p), C.byref(fst), None,
                                                 C.c_void_p(stream.cuda_stream)))
            torch.cuda.synchronize()
            gather_ok = bool(torch.equal(whole.reshape(H, fmt.pitch), full))
    else:
        host = torch.empty(frame_bytes, dtype=torch.uint8).pin_memory()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(5):
            host.copy_(fb[i % F], non_blocking=True)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) / 5 * 1e3

    torch.cuda.synchronize()
    L.nt_camera_table_destroy(C.c_void_p(tab))
    # ---- BASELINE.json configs[4], the config north_star assigns to the 8-GPU split: BoxScene(10) 4096 x 4096, tiled over the
    # ranks in row bands, RCCL gather verified (every rank takes part; reported next to the headline at every N)
    cfg5 = None
    if not args.headline_only:
        try:
            cfg5 = config5(torch, dist, ntracer_amd, tracern, _lib, ntd, rank, world, local_rank)
        except Exception as e:           # never hide the headline
            cfg5 = {"error": repr(e)[:300]}
    # the last collective is behind us: every rank leaves the process group together, here -- rank 0 goes on alone with the CPU
    # baseline and the line (a rank that tears its communicator down long after its peers have exited is how jobs hang)
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        return

    # correctness guard inside the bench: the last rendered frame's checksum equals a fresh single render
    ms_per_step = wall * 1e3 / args.steps
    kernel_us = dev_ms * 1e3 / launches                       # average launch duration (HIP events)
    total_bytes = float(own_rows) * W * fmt.bytes_per_pixel * F * args.steps     # algorithmic: framebuffer write only
    achieved = total_bytes / (dev_ms * 1e-3) / 1e9
    overlap_note = None
    if issue != "one stream":
        # `value` comes from a leg whose launches overlap (two streams) or take another path (camera table): the HIP events above
        # timed the one-stream leg.  The roofline follows `value`: algorithmic bytes of a step over that leg's time per step
        # (launches that overlap have no duration of their own); the one-stream launch stays in `one_stream_launch_us`.
        overlap_note = "N > 1: `value` is the '%s' leg; achieved = algorithmic bytes of a step / that leg's time per step" % issue
        achieved = float(own_rows) * W * fmt.bytes_per_pixel * F / (ms_per_step * 1e-3) / 1e9
    # HBM traffic and instruction counts per call from the committed rocprofv3 PMC passes of this build (separate runs:
    # counters cannot be sampled live)
    traffic = None
    valu = None
    prof = load_profile()
    if prof and world == 1 and prof.get("headline_call", {}).get("frames_per_call") == F:
        hc = prof["headline_call"]
        traffic = hc["write_bytes_per_call"] + hc["fetch_bytes_per_call_corrected"]
        valu = valu_bound(hc, kernel_us)
    elif prof and world == 8 and F == 160 and prof.get("band8_call", {}).get("rows") == own_rows:
        # a rank's launch of an 8-GPU step was profiled too (on one GPU: tools/band_proxy.py --world 8)
        # (two-stream legs: the launch shape nt_render_opts.overlapped selects, profiled one call at a time)
        hc = prof["band8_overlapped_call"] if ("two streams" in issue and "band8_overlapped_call" in prof) else prof["band8_call"]
        traffic = hc["write_bytes_per_call"] + hc["fetch_bytes_per_call_corrected"]
        valu = valu_bound(hc, kernel_us if issue == "one stream" else ms_per_step * 1e3)
    out = {
        "metric": "Mrays/s (primary+shadow), 6-D hypercube @1920x1080",
        "value": round(value, 1), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BoxScene(6) 1920x1080 RGBX8, 160-frame RotatingCamera sequence (configs[2])",
                   "rays_per_step": W * H * F, "frames_per_step": F, "shadow_rays": 0, "launches": launches,
                   "host_issue_ms_per_step": round(issue_s * 1e3 / args.steps, 5),
                   "settle_ms": args.settle_ms,
                   "tiling": ("%d-row bands round-robin over ranks" % band_rows) if world > 1 else "single GPU",
                   "issue": issue,
                   "framebuffer": "resident in HBM (one buffer per frame)",
                   "cameras": "resident in HBM (nt_camera_table_create before the timed region; one kernel launch a step)"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "valu": valu,
                     # what kernels that only store reach on this part in the tile kernel's pattern (measured once with
                     # tools/micro/store_rate.hip, recorded in profiles/README.md): the practical ceiling of this workload
                     "store_only": {"GBs": STORE_ONLY_GBS, "frac": round(achieved / STORE_ONLY_GBS, 4), "source": "tools/micro/store_rate.hip"},
                     "algorithmic_bytes_per_launch": float(own_rows) * W * fmt.bytes_per_pixel * F,
                     "kernel": "box_tile_kernel<6, false, ROWS, WAVES> (64, 1 for the full frame); one nt_render_table_device call = this kernel "
                               "and nothing else: up to eight dimensions it needs no box_redo_kernel after it",
                     "avg_launch_us": round(kernel_us if overlap_note is None else ms_per_step * 1e3, 2),
                     "one_stream_launch_us": round(kernel_us, 2), "issue_note": overlap_note,
                     "algorithmic_bytes_per_ray": fmt.bytes_per_pixel,
                     "note": "BoxScene reads no scene memory: the only algorithmic HBM traffic is the packed framebuffer "
                             "(4 B/ray); the kernel is bound by instruction issue and latency, not by HBM (see DESIGN.md 4.1), so the HBM "
                             "fraction is structurally small -- `valu` is the bound that binds"},
        "one_stream": {"what": "the K steps on one stream, one kernel at a time, cameras resident (= `value` at N = 1)", "ms_per_step": round(one_stream_ms, 5),
                       "value": round(one_stream_value, 1)},
        "two_streams": None if two_ms is None else {
            "what": "the K steps issued alternately on two streams (two scene handles, two sets of framebuffers, nt_render_opts.overlapped; cameras resident): "
                    "consecutive calls overlap (= `value` at N > 1 when it is the faster way)",
            "ms_per_step": round(two_ms, 5), "value": round(float(W) * H * F / (two_ms * 1e-3) / 1e6, 1)},
        "cold_start": {"what": "the first %d steps after %d warm-up steps from an idle chip (no settling: the clock transient of the first ~50 ms of load)"
                               % (args.steps, args.warmup), "ms_per_step": round(cold_ms, 5), "value": round(float(W) * H * F / (cold_ms * 1e-3) / 1e6, 1)},
        "host_cameras": None if host_ms is None else {
            "what": "the same steps with the cameras handed over as host arrays on every call (nt_render_frames_device: packed, uploaded and "
                    "launched each time -- the PCIe-inclusive rate; the reference's boundary sets a camera per frame): never `value`",
            "ms_per_step": round(host_ms, 5), "value": round(float(W) * H * F / (host_ms * 1e-3) / 1e6, 1),
            "two_streams_ms_per_step": round(host_two_ms, 5), "two_streams_value": round(float(W) * H * F / (host_two_ms * 1e-3) / 1e6, 1)},
        "delivery": {"what": "RCCL gather to rank 0" if world > 1 else "D2H copy to pinned host memory",
                     "ms_per_frame": round(gather_ms, 4), "verified_equal_to_single_gpu_frame": gather_ok,
                     "value_incl_delivery": round(float(W) * H * F / ((ms_per_step + gather_ms * F) * 1e-3) / 1e6, 1)},
    }

    if cfg5 is not None:
        out["config5"] = cfg5
    if args.headline_only:
        args.no_cpu_baseline = args.no_extra = True
    if world == 1 and not args.headline_only:
        try:
            out["value_rgbf32"] = value_rgbf32(torch, ntracer_amd, tracern, _lib, origins, axes, F)
            out["scaling_proxy"] = scaling_proxy(torch, ntracer_amd, tracern, _lib, ntd, origins, axes, F, ms_per_step, two_ms)
        except Exception as e:       # never hide the headline
            out["scaling_proxy"] = {"error": repr(e)}
    if world == 1 and not args.headline_only:
        try:
            out["dropin_render"] = dropin_render(torch, ntracer_amd, tracern)
        except Exception as e:
            out["dropin_render"] = {"error": repr(e)[:300]}
    if not args.no_cpu_baseline:
        # (rank 0 only, after every timed region; at N > 1 the other ranks have left by now)
        out["cpu_baseline"] = cpu_baseline(origins, axes, W, H)
    if not args.no_extra and world == 1:
        try:
            out["extra"] = extra_configs(torch, ntracer_amd, tracern, _lib)
        except Exception as e:       # extras never hide the headline
            out["extra"] = {"error": repr(e)}
    emit(out)
    if dist is not None:
        dist.destroy_process_group()


def compact(out):
    """The ONE line for stdout: every key of the bench contract -- `roofline` and `cpu_baseline` whole in what they claim,
    bare of prose -- plus the headline figures of everything else bench.py measures.  The full record (every leg with its
    description, instruction counts by class, the CPU samples, the drop-in call, the extras) goes to stderr and to
    gpurun_out/bench_details.json: at ten kilobytes it no longer fits a log tail."""
    def pick(d, keys):
        return None if not isinstance(d, dict) else {k: d[k] for k in keys if k in d}
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c["config"] = pick(out["config"], ("workload", "rays_per_step", "frames_per_step", "shadow_rays", "launches", "tiling", "issue", "settle_ms"))
    c["config"]["inputs"] = "cameras and framebuffers resident in HBM; one kernel launch a step"
    r = out["roofline"]
    c["roofline"] = pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "algorithmic_bytes_per_launch", "algorithmic_bytes_per_ray",
                             "one_stream_launch_us"))
    c["roofline"]["kernel"] = "box_tile_kernel<6, false, ROWS, WAVES>"
    c["roofline"]["store_only_frac"] = (r.get("store_only") or {}).get("frac")
    c["roofline"]["valu"] = pick(r.get("valu"), ("frac_lo", "frac", "frac_hi", "valu_wave_insts", "kernel_cycles_profiled", "source"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind"))
        c["cpu_baseline"]["sample"] = (cb.get("sample") or "").split(" (")[0]          # (without the parenthesis on the thread pool)
        c["cpu_baseline"]["cell120"] = pick(cb.get("cell120"), ("value", "unit", "cores", "kind"))
        pv = cb.get("port_vs_reference")
        if isinstance(pv, dict):
            c["cpu_baseline"]["port_over_reference"] = {k[:-6] if k.endswith("_1080p") else k: v.get("port_over_reference") for k, v in pv.items()
                                                        if isinstance(v, dict) and "port_over_reference" in v}
    for k in ("one_stream", "two_streams", "cold_start", "host_cameras"):
        c[k] = pick(out.get(k), ("ms_per_step", "value", "two_streams_ms_per_step", "two_streams_value"))
    c["delivery"] = pick(out.get("delivery"), ("ms_per_frame", "verified_equal_to_single_gpu_frame"))
    c["config5"] = pick(out.get("config5"), ("value", "unit", "n_gpus", "ms_per_step", "frames_per_step", "issue", "one_stream_ms_per_step", "two_streams_ms_per_step",
                                             "gather_verified_equal_to_whole_frame", "error"))
    c["value_rgbf32"] = pick(out.get("value_rgbf32"), ("value", "ms_per_step", "hbm_write_GBs", "hbm_frac"))
    c["scaling_proxy"] = pick(out.get("scaling_proxy"), ("ms_per_step", "two_streams_ms_per_step", "full_step_ms", "two_streams_vs_one_stream_full_step", "measured_on", "error"))
    dr = out.get("dropin_render")
    if isinstance(dr, dict):
        c["dropin_render_ms_per_frame"] = {k: v.get("ms_per_frame") for k, v in dr.items() if isinstance(v, dict) and "ms_per_frame" in v}
    ex = out.get("extra")
    if isinstance(ex, dict):
        c["extra"] = {k: ex[k] for k in ("cfg1_box3_1080p_Mrays_s", "cfg3_cell120_1080p_Mrays_s", "cfg3_ms_per_frame", "cfg3_strict_reference_ms_per_frame") if k in ex}
        sh = ex.get("cfg3_shadows_1080p")
        if isinstance(sh, dict):
            c["extra"]["cfg3_shadows_1080p_Mrays_s_primary_plus_shadow"] = sh.get("Mrays_s_primary_plus_shadow")
    c["details"] = "stderr and gpurun_out/bench_details.json"
    return c


def emit(out):
    full = json.dumps(out)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_details.json"), "w") as fh:
            fh.write(full + "\n")
    except OSError:
        pass
    sys.stderr.write("# bench.py details: " + full + "\n")
    sys.stderr.flush()
    print(json.dumps(compact(out)), flush=True)


def self_launch(gpus):
    """One process per GPU through torch.distributed.run (the launcher the driver uses), rendezvous on 127.0.0.1."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """The N-rank plumbing without a GPU: rendezvous (gloo), the band split bench.py uses, barrier-bracketed timing with the MAX
    over ranks, the gather of every rank's compact band buffer to rank 0 and its check against the whole frame.  The band
    buffers hold a synthetic pattern (a function of row and byte position), not rendered pixels: nothing here stands in for
    the HIP path, and the line says so (`dry_run`, value null)."""
    import torch
    import torch.distributed as dist
    import ntracer_amd
    from ntracer_amd import distributed as ntd
    W, H = 1920, 1080
    fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in RGBX8])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    band_rows = pick_band_rows(ntd, H, world)
    rows = ntd.owned_rows(H, rank, world, band_rows)

    def pattern(ys):
        y = torch.as_tensor(np.asarray(ys), dtype=torch.int64).reshape(-1, 1)
        x = torch.arange(fmt.pitch, dtype=torch.int64).reshape(1, -1)
        return ((y * 131 + x * 7 + (y * x) % 251) % 256).to(torch.uint8)

    compact = pattern(rows)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.barrier()
    wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    full = ntd.gather_framebuffer(compact, fmt, rank, world, dst=0, band_rows=band_rows) if world > 1 else compact
    if rank == 0:
        ok = bool(torch.equal(full, pattern(range(H))))
        print(json.dumps({"metric": "Mrays/s (primary+shadow), 6-D hypercube @1920x1080", "value": None, "unit": "Mrays/s", "dry_run": True,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "band_rows": band_rows,
                          "rows_per_rank": [int(len(ntd.owned_rows(H, r, world, band_rows))) for r in range(world)],
                          "gather_verified": ok, "barrier_loop_s": round(float(wall.item()), 4)}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def pick_band_rows(ntd, H, world):
    """band height: the largest of 32 / 16 / 8 rows that leaves the busiest rank the fewest rows"""
    return min((32, 16, 8), key=lambda b: (max(len(ntd.owned_rows(H, r, world, b)) for r in range(world)), -b))


SETTLE_MS = 200.0


def settle(torch, fn, ms):
    """Issue `fn` back to back for at least `ms` of wall time (a few calls deep in the queue), untimed: every timed region starts
    on a chip that has been under this load long enough to have settled its clock."""
    if ms <= 0:
        return
    t0 = time.perf_counter()
    k = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        fn()
        k += 1
        if k % 8 == 0:
            torch.cuda.synchronize()


def _time_frames(torch, _lib, scene, fmt, origins, axes, frames, reps, opts=None, rows=None):
    """ms per call of nt_render_frames_device (HIP events on the launch stream), framebuffers resident"""
    fst = fmt._as_struct()
    frame_bytes = (rows if rows is not None else fmt.height) * fmt.pitch
    fb = torch.empty((frames, frame_bytes), dtype=torch.uint8, device="cuda")
    o = np.ascontiguousarray(origins[:frames], np.float32)
    a = np.ascontiguousarray(axes[:frames], np.float32)
    st = torch.cuda.current_stream()

    def go():
        _lib.check(_lib.lib().nt_render_frames_device(scene._handle, C.c_void_p(fb.data_ptr()), frame_bytes, frames, o.ctypes.data_as(_lib.f32p),
                                                      a.ctypes.data_as(_lib.f32p), C.byref(fst), C.byref(opts) if opts is not None else None,
                                                      C.c_void_p(st.cuda_stream)))
    for _ in range(3):
        go()
    settle(torch, go, SETTLE_MS)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        go()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def value_rgbf32(torch, ntracer_amd, tracern, _lib, origins, axes, F):
    """The headline workload into three fp32 channels (12 B/ray, big-endian floats: SURVEY 8d's second format): the stored
    values are the reference's colours themselves, x / sqrtf(sq) computed as it stands."""
    fmt = ntracer_amd.ImageFormat(1920, 1080, [ntracer_amd.Channel(*c) for c in RGBF32])
    ms = _time_frames(torch, _lib, tracern.BoxScene(6), fmt, origins, axes, F, 10)
    rays = 1920 * 1080 * F
    prof = load_profile()
    return {"value": round(rays / ms / 1e3, 1), "unit": "Mrays/s", "ms_per_step": round(ms, 5), "bytes_per_ray": 12,
            "hbm_write_GBs": round(rays * 12 / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(rays * 12 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "valu": valu_bound(prof.get("rgbf32_call"), ms * 1e3) if prof and F == 160 else None,
            "workload": "BoxScene(6) 1920x1080, three fp32 channels, the same %d-frame sequence, one call per step" % F}


def scaling_proxy(torch, ntracer_amd, tracern, _lib, ntd, origins, axes, F, ms_full, ms_full_two=None):
    """What ONE rank of `--gpus 8` does per step, timed on this GPU: rank 0's bands (8 rows each, dealt round-robin to 8
    ranks) of every frame, compact buffer, issued the ways `bench.py --gpus 8` issues them (cameras resident: one stream; two
    streams with nt_render_opts.overlapped) and with host cameras.  No 8-GPU run is behind these numbers; they bound the
    strong-scaling factor the kernels allow (full step / this), before any inter-GPU effect."""
    fmt = ntracer_amd.ImageFormat(1920, 1080, [ntracer_amd.Channel(*c) for c in RGBX8])
    fst = fmt._as_struct()
    opts = _lib.NtRenderOpts()
    opts.device = torch.cuda.current_device()
    opts.band_rank, opts.band_world, opts.band_rows, opts.compact = 0, 8, 8, 1
    opts2 = _lib.NtRenderOpts()
    C.memmove(C.byref(opts2), C.byref(opts), C.sizeof(opts))
    opts2.overlapped = 1
    rows = len(ntd.owned_rows(1080, 0, 8, 8))
    L = _lib.lib()
    o = np.ascontiguousarray(origins[:F], np.float32)
    a = np.ascontiguousarray(axes[:F], np.float32)
    tab = L.nt_camera_table_create(6, F, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), torch.cuda.current_device())
    if not tab:
        return {"error": _lib.last_error()}
    scs = [tracern.BoxScene(6), tracern.BoxScene(6)]
    fbs = [torch.empty((F, rows * fmt.pitch), dtype=torch.uint8, device="cuda") for _ in range(2)]
    sts = [torch.cuda.current_stream(), torch.cuda.Stream()]
    state = {"k": 0}

    def table_call(i, op):
        return (scs[i]._handle, C.c_void_p(fbs[i].data_ptr()), rows * fmt.pitch, C.c_void_p(tab), 0, F, C.byref(fst), C.byref(op), C.c_void_p(sts[i].cuda_stream))

    def host_call(i, op):
        return (scs[i]._handle, C.c_void_p(fbs[i].data_ptr()), rows * fmt.pitch, F, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(fst), C.byref(op),
                C.c_void_p(sts[i].cuda_stream))

    def leg(fn, calls, steps=80):
        def go():
            _lib.check(fn(*calls[state["k"] % len(calls)]))
            state["k"] += 1
        for _ in range(6):
            go()
        settle(torch, go, SETTLE_MS)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            go()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / steps
    ms = leg(L.nt_render_table_device, [table_call(0, opts)])
    ms2 = leg(L.nt_render_table_device, [table_call(0, opts2), table_call(1, opts2)])
    hms = leg(L.nt_render_frames_device, [host_call(0, opts)])
    hms2 = leg(L.nt_render_frames_device, [host_call(0, opts2), host_call(1, opts2)])
    torch.cuda.synchronize()
    L.nt_camera_table_destroy(C.c_void_p(tab))
    out = {"what": "rank 0 of 8: bands of 8 rows, %d of 1080 rows of each of the %d frames, on one GPU" % (rows, F),
           "ms_per_step": round(ms, 5), "full_step_ms": round(ms_full, 5), "implied_speedup_at_8": round(ms_full / ms, 2),
           "ideal_ms": round(ms_full / 8, 5), "measured_on": "1 GPU (no 8-GPU node was available to the builder)",
           # (against the full step issued the same way, and against `value`'s -- one stream -- which is what a scaling curve divides by)
           "two_streams_ms_per_step": round(ms2, 5), "two_streams_vs_one_stream_full_step": round(ms_full / ms2, 2),
           "host_cameras_ms_per_step": round(hms, 5), "host_cameras_two_streams_ms_per_step": round(hms2, 5)}
    if ms_full_two:
        out["two_streams_full_step_ms"] = round(ms_full_two, 5)
        out["two_streams_implied_speedup_at_8"] = round(ms_full_two / ms2, 2)
    return out


def config5(torch, dist, ntracer_amd, tracern, _lib, ntd, rank, world, local_rank, frames=16, steps=8, warmup=2):
    """BASELINE.json configs[4]: BoxScene(10) (the reference sends n = 10 through its generic var_geometry module; here a
    compile-time-N kernel), 4096 x 4096 RGBX8, `frames` cameras of the rotation per call -- resident in device memory, as the
    headline's -- every frame tiled over the ranks in row bands (band b -> rank b % N, as the headline), barrier-bracketed, MAX
    over ranks; then one frame gathered to rank 0 (RCCL) and compared with the same frame rendered whole."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "box_n10_4096x4096.npz"))
    n, W, H = 10, 4096, 4096
    sel = (np.arange(frames) * (len(g["origins"]) // frames)) % len(g["origins"])            # spread over the rotation
    o = np.ascontiguousarray(g["origins"][sel], np.float32)
    a = np.ascontiguousarray(g["axes"][sel], np.float32)
    fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in RGBX8])
    fst = fmt._as_struct()
    band_rows = pick_band_rows(ntd, H, world)
    opts = _lib.NtRenderOpts()
    opts.device = local_rank
    opts.band_rank, opts.band_world, opts.band_rows, opts.compact = rank, world, band_rows, 1
    opts2 = _lib.NtRenderOpts()
    C.memmove(C.byref(opts2), C.byref(opts), C.sizeof(opts))
    opts2.overlapped = 1
    own = len(ntd.owned_rows(H, rank, world, band_rows))
    L = _lib.lib()
    tab = L.nt_camera_table_create(n, frames, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), local_rank)
    if not tab:
        return {"error": _lib.last_error()}
    scs = [tracern.BoxScene(n), tracern.BoxScene(n)]
    fbs = [torch.empty((frames, own * fmt.pitch), dtype=torch.uint8, device="cuda") for _ in range(2)]
    sts = [torch.cuda.current_stream(), torch.cuda.Stream()]
    fb, sc, st = fbs[0], scs[0], sts[0]
    state = {"k": 0}

    def table_call(i, op):
        return (scs[i]._handle, C.c_void_p(fbs[i].data_ptr()), own * fmt.pitch, C.c_void_p(tab), 0, frames, C.byref(fst), C.byref(op), C.c_void_p(sts[i].cuda_stream))

    def host_call(i, op):
        return (scs[i]._handle, C.c_void_p(fbs[i].data_ptr()), own * fmt.pitch, frames, o.ctypes.data_as(_lib.f32p), a.ctypes.data_as(_lib.f32p), C.byref(fst),
                C.byref(op), C.c_void_p(sts[i].cuda_stream))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def leg(fn, calls):
        """EXACTLY `steps` steps between two barriers after warm-up and settling; MAX over ranks; ms per step."""
        def go():
            _lib.check(fn(*calls[state["k"] % len(calls)]))
            state["k"] += 1
        for _ in range(warmup * len(calls)):
            go()
        settle(torch, go, SETTLE_MS)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            go()
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()) * 1e3 / steps
    one_ms = leg(L.nt_render_table_device, [table_call(0, opts)])
    # ... and alternately on two streams (two scene handles, two sets of framebuffers, nt_render_opts.overlapped), as the headline does
    two_ms = leg(L.nt_render_table_device, [table_call(0, opts2), table_call(1, opts2)])
    # ... and with the cameras as host arrays on every call (the PCIe-inclusive way; never `value`)
    host_ms = leg(L.nt_render_frames_device, [host_call(0, opts)])
    host_two_ms = leg(L.nt_render_frames_device, [host_call(0, opts2), host_call(1, opts2)])
    two_wins = world > 1 and two_ms < one_ms
    ms = two_ms if two_wins else one_ms
    # leave fb holding the frames of a one-stream call (the gather below looks at the last one)
    _lib.check(L.nt_render_table_device(*table_call(0, opts)))
    torch.cuda.synchronize()
    ok = None
    gather_ms = None
    if dist is not None:
        t1 = time.perf_counter()
        full = ntd.gather_framebuffer(fb[frames - 1], fmt, rank, world, dst=0, band_rows=band_rows)
        torch.cuda.synchronize()
        dist.barrier()
        gather_ms = (time.perf_counter() - t1) * 1e3
        if rank == 0:
            whole = torch.empty(H * fmt.pitch, dtype=torch.uint8, device="cuda")
            _lib.check(L.nt_render_frames_device(sc._handle, C.c_void_p(whole.data_ptr()), H * fmt.pitch, 1, o[frames - 1:].ctypes.data_as(_lib.f32p),
                                                 a[frames - 1:].ctypes.data_as(_lib.f32p), C.byref(fst), None, C.c_void_p(st.cuda_stream)))
            torch.cuda.synchronize()
            ok = bool(torch.equal(whole.reshape(H, fmt.pitch), full))
    torch.cuda.synchronize()
    L.nt_camera_table_destroy(C.c_void_p(tab))
    rays = float(W) * H * frames
    return {"workload": "BoxScene(10) 4096x4096 RGBX8, %d cameras of the rotation per call (configs[4]), cameras resident in device memory" % frames,
            "value": round(rays / (ms * 1e-3) / 1e6, 1),
            "unit": "Mrays/s", "n_gpus": world, "ms_per_step": round(ms, 4), "steps": steps, "frames_per_step": frames,
            "tiling": ("%d-row bands round-robin over ranks" % band_rows) if world > 1 else "single GPU", "scaling": "strong",
            "issue": "steps alternate between two streams" if two_wins else "one stream", "one_stream_ms_per_step": round(one_ms, 4),
            "two_streams_ms_per_step": round(two_ms, 4),
            "host_cameras": {"one_stream_ms_per_step": round(host_ms, 4), "two_streams_ms_per_step": round(host_two_ms, 4),
                             "value": round(rays / (host_ms * 1e-3) / 1e6, 1)},
            "gather_ms_per_frame": None if gather_ms is None else round(gather_ms, 3), "gather_verified_equal_to_whole_frame": ok}


def dropin_render(torch, ntracer_amd, tracern):
    """The call the reference actually makes (obj_BlockingRenderer_render, src/render.cpp:853-909): ONE frame into a HOST buffer --
    `BlockingRenderer().render(bytearray, format, scene)` at 1920x1080 RGBX8 -- kernel + D2H + every synchronisation, wall clock
    per call; for BoxScene(6) and the 120-cell, cameras of the rotation set through the scene API as a render loop would.
    `pipelined`: two scenes and two renderers on two host threads, frame k + 1 rendering while frame k's copy runs."""
    import threading
    W, H = 1920, 1080
    fmt = ntracer_amd.ImageFormat(W, H, [ntracer_amd.Channel(*c) for c in RGBX8])
    G = os.path.join(ROOT, "tests", "golden")
    res = {"what": "BlockingRenderer().render(bytearray, ImageFormat(1920, 1080, RGBX8), scene): one frame into host memory per call (kernel + D2H + syncs)"}

    def measure(make_scene, cams_o, cams_a, frames, key):
        sc = make_scene()
        r = ntracer_amd.BlockingRenderer()
        buf = bytearray(fmt.pitch * H)
        for k in range(3):
            sc._set_camera_arrays(cams_o[k], cams_a[k])
            r.render(buf, fmt, sc)
        t = []
        for k in range(frames):
            sc._set_camera_arrays(cams_o[k % len(cams_o)], cams_a[k % len(cams_o)])
            t0 = time.perf_counter()
            r.render(buf, fmt, sc)
            t.append(time.perf_counter() - t0)
        ms = float(np.median(t)) * 1e3
        out = {"ms_per_frame": round(ms, 4), "best_ms": round(min(t) * 1e3, 4), "Mrays_s": round(W * H / ms / 1e3, 1), "frames": frames}
        # pipelined: two scene handles (each has its own stream and device framebuffer), two threads, alternate frames
        scs = [make_scene(), make_scene()]
        rs = [ntracer_amd.BlockingRenderer(), ntracer_amd.BlockingRenderer()]
        bufs = [bytearray(fmt.pitch * H), bytearray(fmt.pitch * H)]

        def worker(i, count):
            for k in range(count):
                f = (2 * k + i) % len(cams_o)
                scs[i]._set_camera_arrays(cams_o[f], cams_a[f])
                rs[i].render(bufs[i], fmt, scs[i])
        for i in (0, 1):
            worker(i, 2)
        th = [threading.Thread(target=worker, args=(i, frames // 2)) for i in (0, 1)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        ms2 = (time.perf_counter() - t0) * 1e3 / (2 * (frames // 2))
        out["pipelined_ms_per_frame"] = round(ms2, 4)
        out["pipelined_Mrays_s"] = round(W * H / ms2 / 1e3, 1)
        res[key] = out

    g = np.load(os.path.join(G, "box_n6_1920x1080.npz"))
    measure(lambda: tracern.BoxScene(6), g["origins"], g["axes"], 60, "box6")
    g4 = np.load(os.path.join(G, "cell120_n4.npz"))
    measure(lambda: tracern.CompositeScene.from_flat(4, g4), g4["origins"], g4["axes"], 20, "cell120")
    # the D2H copy of one such frame alone, to pinned memory (what the call cannot be faster than)
    dev = torch.empty(fmt.pitch * H, dtype=torch.uint8, device="cuda")
    host = torch.empty(fmt.pitch * H, dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()
    t = []
    for _ in range(10):
        t0 = time.perf_counter()
        host.copy_(dev, non_blocking=True)
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    res["d2h_alone_ms"] = round(min(t) * 1e3, 4)
    return res


def cpu_quota_cores():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota if there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(origins, axes, W, H):
    """The oracle (a port of the reference's path: the same 32x32 chunk queue; hardware_concurrency()-1 worker threads
    that persist between frames and sleep on a condition variable, plus the caller -- render.cpp:769-909) on a bounded
    sample of the same frames, all inside ONE C call (no Python between frames).  Where a cgroup quota gives the process
    fewer CPUs than the machine has, a second sample with that many threads is reported beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    def sample(threads, seconds):
        r = ob.OracleRenderer(threads)
        sc = ob.OracleScene(6, origins[0], axes[0])
        r.render_frames(sc, W, H, RGBX8, origins, axes, 3)          # warm-up: threads started, pages touched
        t0 = time.perf_counter()
        _, secs = r.render_frames(sc, W, H, RGBX8, origins, axes, 1600, max_seconds=seconds)
        total = time.perf_counter() - t0
        n = r.threads
        r.close()
        rate = lambda t: W * H / t / 1e6
        return {"value": round(W * H * len(secs) / total / 1e6, 2), "cores": n, "frames": len(secs), "seconds": round(total, 1),
                "best_frame_Mrays_s": round(rate(float(secs.min())), 1), "median_frame_Mrays_s": round(rate(float(np.median(secs))), 1)}

    online = os.cpu_count() or 1
    quota = cpu_quota_cores()

    def describe(a, what):
        return "%d consecutive frames of the same 1920x1080 BoxScene(6) rotation in one C call, %.1f s, %d threads (persistent pool: %d workers + " \
               "the caller%s)" % (a["frames"], a["seconds"], a["cores"], a["cores"] - 1, what)

    if quota >= online:
        a = sample(-1, 12.0)               # threads = hardware_concurrency() - 1 + the caller, as BlockingRenderer()
        out = dict(a, unit="Mrays/s", kind="port", sample=describe(a, ", the reference's default"))
    else:
        # a cgroup quota below the CPU count: the reference's default thread count oversubscribes it.  `value` is the sample with
        # as many threads as CPUs the process may use; the default-thread-count sample is reported beside it.
        a = sample(quota - 1, 7.0)
        out = dict(a, unit="Mrays/s", kind="port", sample=describe(a, "; = the CPUs the cgroup lets this process use"))
        b = sample(-1, 7.0)
        out["reference_default_threads"] = dict(b, sample=describe(b, ", the reference's default, on %d usable CPUs" % quota))
    out["cpus_online"] = online
    out["cpus_usable"] = quota
    for k in ("frames", "seconds"):
        out.pop(k, None)
    # ---- the 120-cell (configs[3]) on the same cores: the reference's k-d walk restated (tracer.hpp:1179-1243 under
    # render.cpp:829-838's thread rule), reference tree, cameras of the rotation, bounded to ~8 s
    try:
        g4 = np.load(os.path.join(ROOT, "tests", "golden", "cell120_n4.npz"))
        flat = {k: g4[k] for k in ("root", "node_axis", "node_split", "node_left", "node_right", "items", "batch_recs", "batch_mats", "tri_recs",
                                   "tri_mats", "solid_recs", "solid_types", "solid_mats", "materials", "aabb_start", "aabb_end")}
        flat["batch_size"] = 4
        sel = [0, 20, 40, 60, 80, 100, 120, 140]                    # the cameras the GPU figure (extra.cfg3_*) is timed on
        r = ob.OracleRenderer(min(quota, online) - 1)
        sc = ob.OracleScene(4, g4["origins"][0], g4["axes"][0], flat=flat)
        t0 = time.perf_counter()
        _, secs = r.render_frames(sc, W, H, RGBX8, g4["origins"][sel], g4["axes"][sel], 64, max_seconds=8.0)
        total = time.perf_counter() - t0
        out["cell120"] = {"value": round(W * H * len(secs) / total / 1e6, 3), "unit": "Mrays/s", "cores": r.threads, "kind": "port",
                          "best_frame_Mrays_s": round(W * H / float(secs.min()) / 1e6, 3),
                          "sample": "%d frames of the 1920x1080 {5/2,3,3} 120-cell (reference-built tree, the reference's walk) in one C call, %.1f s, %d threads"
                                    % (len(secs), total, r.threads)}
        r.close()
    except Exception as e:          # a baseline never hides the headline
        out["cell120"] = {"error": repr(e)[:200]}
    # ---- how the port compares with the reference itself (measured where the reference can run: the build container)
    try:
        out["port_vs_reference"] = json.load(open(os.path.join(ROOT, "profiles", "port_vs_reference.json")))
    except (OSError, ValueError):
        out["port_vs_reference"] = None
    return out


def extra_configs(torch, ntracer_amd, tracern, _lib):
    """Other BASELINE.json configs (cfgK = configs[K]), device-resident timing (not the headline)."""
    res = {}
    G = os.path.join(ROOT, "tests", "golden")
    prof = load_profile()

    def time_scene(scene, fmt, origins, axes, frames, reps, strict=False):
        ropts = _lib.NtRenderOpts()
        ropts.device = -1
        ropts.band_world = 1
        ropts.strict_reference = 1 if strict else 0
        return _time_frames(torch, _lib, scene, fmt, origins, axes, frames, reps, opts=ropts) / frames      # ms per frame

    chan = [ntracer_amd.Channel(*c) for c in RGBX8]
    g = np.load(os.path.join(G, "box_n3_1920x1080.npz"))
    ms = time_scene(tracern.BoxScene(3), ntracer_amd.ImageFormat(1920, 1080, chan), g["origins"], g["axes"], 160, 10)
    res["cfg1_box3_1080p_Mrays_s"] = round(1920 * 1080 / ms / 1e3, 1)
    g = np.load(os.path.join(G, "box_n10_4096x4096.npz"))
    ms = time_scene(tracern.BoxScene(10), ntracer_amd.ImageFormat(4096, 4096, chan), g["origins"], g["axes"], 4, 10)
    res["cfg4_box10_4096_Mrays_s"] = round(4096 * 4096 / ms / 1e3, 1)
    g = np.load(os.path.join(G, "cell120_n4.npz"))
    sc = tracern.CompositeScene.from_flat(4, g)
    sel = [0, 20, 40, 60, 80, 100, 120, 140]
    ms = time_scene(sc, ntracer_amd.ImageFormat(1920, 1080, chan), g["origins"][sel], g["axes"][sel], 8, 2)
    res["cfg3_cell120_1080p_Mrays_s"] = round(1920 * 1080 / ms / 1e3, 1)
    res["cfg3_ms_per_frame"] = round(ms, 3)
    if prof and "config4_call" in prof:
        # the packet kernel's bound is instruction issue too (the scene is L2-resident: see DESIGN.md 4.2)
        res["cfg3_valu"] = valu_bound(prof["config4_call"], ms * 1e3 * prof["config4_call"].get("frames_per_call", 8))
    # the same frames walking exactly the cells the reference walks (nt_render_opts.strict_reference; same bytes)
    ms_strict = time_scene(sc, ntracer_amd.ImageFormat(1920, 1080, chan), g["origins"][sel], g["axes"][sel], 8, 2, strict=True)
    res["cfg3_strict_reference_ms_per_frame"] = round(ms_strict, 3)
    # SURVEY 8d byte model on the reference tree, oracle counters on frame 0: the reference's walk (32.1 branches,
    # 4.68 leaves, 195 simplices per ray) and the default walk that drops cells beyond the hit (31.2 / 4.46 / 168)
    bytes_per_ray = 16 * 32.1 + 8 * 4.68 + 195 * (4 + 4 * 21) + 4
    res["cfg3_algorithmic_bytes_per_ray"] = round(bytes_per_ray)
    res["cfg3_algorithmic_TB_s"] = round(bytes_per_ray * 1920 * 1080 / (ms_strict * 1e-3) / 1e12, 2)
    bytes_pruned = 16 * 31.2 + 8 * 4.46 + 168 * (4 + 4 * 21) + 4
    res["cfg3_default_walk_bytes_per_ray"] = round(bytes_pruned)
    res["cfg3_default_walk_TB_s"] = round(bytes_pruned * 1920 * 1080 / (ms * 1e-3) / 1e12, 2)
    # the reference's own simplices and batches under OUR k-d tree (nt_kdtree_build): identical pixels, fewer tests
    try:
        t0 = time.perf_counter()
        reb = sc.with_rebuilt_tree()
        reb_s = time.perf_counter() - t0
        ms_reb = time_scene(reb, ntracer_amd.ImageFormat(1920, 1080, chan), g["origins"][sel], g["axes"][sel], 8, 2)
        res["cfg3_rebuilt_tree"] = {"ms_per_frame": round(ms_reb, 3), "Mrays_s": round(1920 * 1080 / ms_reb / 1e3, 1),
                                       "build_s": round(reb_s, 1), "nodes": int(len(reb._flat["node_axis"]))}
    except Exception as e:
        res["cfg3_rebuilt_tree"] = {"error": str(e)[:200]}
    # the same polytope generated and partitioned on our side (ntracer_amd.polytope + nt_kdtree_build) -- SURVEY 8d's
    # "build-tree figure"; pixels equal the reference-built scene's except on silhouettes (the reference inflates facets)
    try:
        from ntracer_amd import polytope
        t0 = time.perf_counter()
        _, own, _ = polytope.build_scene(["5/2", "3", "3"])
        built_s = time.perf_counter() - t0
        ms_own = time_scene(own, ntracer_amd.ImageFormat(1920, 1080, chan), g["origins"][sel], g["axes"][sel], 8, 2)
        res["cfg3_own_scene"] = {"ms_per_frame": round(ms_own, 3), "Mrays_s": round(1920 * 1080 / ms_own / 1e3, 1),
                                    "generate_and_build_s": round(built_s, 1)}
    except Exception as e:          # the headline must not depend on the generator
        res["cfg3_own_scene"] = {"error": str(e)[:200]}
    # the same scene with shadows on, one point light and one global light: primary + shadow rays
    n = 4
    sc.add_light(tracern.PointLight(tracern.Vector(n, (8.0, 9.0, -7.0, 3.0)), (60.0, 60.0, 60.0)))
    sc.add_light(tracern.GlobalLight(tracern.Vector(n, (0.2, -1.0, 0.3, 0.1)).unit(), (0.5, 0.5, 0.5)))
    sc.set_shadows(True)
    w2, h2 = 960, 540
    fmt2 = ntracer_amd.ImageFormat(w2, h2, chan)
    sc._set_camera_arrays(g["origins"][0], g["axes"][0])
    buf = bytearray(fmt2.pitch * h2)
    ntracer_amd.BlockingRenderer().render(buf, fmt2, sc, collect_stats=True)
    st = sc.last_stats()
    ms2 = time_scene(sc, fmt2, g["origins"][sel], g["axes"][sel], 4, 2)
    res["cfg3_shadows_960x540"] = {"primary_rays": st["rays"], "shadow_rays": st["shadow_rays"],
                                      "Mrays_s_primary_plus_shadow": round((st["rays"] + st["shadow_rays"]) / ms2 / 1e3, 1),
                                      "ms_per_frame": round(ms2, 3)}
    # ... and at the headline's size, eight cameras a call (rays counted on frame 0 by a statistics render)
    fmt3 = ntracer_amd.ImageFormat(1920, 1080, chan)
    buf = bytearray(fmt3.pitch * 1080)
    ntracer_amd.BlockingRenderer().render(buf, fmt3, sc, collect_stats=True)
    st = sc.last_stats()
    ms3 = time_scene(sc, fmt3, g["origins"][sel], g["axes"][sel], 8, 2)
    res["cfg3_shadows_1080p"] = {"primary_rays": st["rays"], "shadow_rays": st["shadow_rays"],
                                    "Mrays_s_primary_plus_shadow": round((st["rays"] + st["shadow_rays"]) / ms3 / 1e3, 1),
                                    "ms_per_frame": round(ms3, 3), "frames_per_call": 8}
    if prof and "shadow" in prof:
        res["cfg3_shadows_1080p"]["kernels_profiled"] = {k: {"kernel_cycles": round(v.get("kernel_cycles", 0)), "valu_wave_insts": round(v.get("SQ_INSTS_VALU", 0)),
                                                               "issue_frac_lo": round(v.get("issue_frac_lo", 0), 3), "issue_frac_hi": round(v.get("issue_frac_hi", 0), 3)}
                                                           for k, v in prof["shadow"].items() if k.startswith("composite") and "true, true" not in k}
    return res


if __name__ == "__main__":
    main()

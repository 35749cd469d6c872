#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/profile_round.sh into <tag>_pmc_summary.json (+ readable tables).
python3 tools/pmc_summary.py <raw dir> <out dir> <tag>"""
import collections
import csv
import glob
import json
import os
import re
import sys

raw, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
SIMDS = 256 * 4
CLOCK_GHZ = 2.4


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name).strip()


def newest(files):
    """the newest file only: raw directories merged back from several GPU runs hold one set of files per run"""
    return sorted(files, key=os.path.getmtime)[-1:]


def collect(d):
    """{kernel: {counter: [per-dispatch sums]}}: a dispatch's counter value is the sum of its rows"""
    files = newest(glob.glob(os.path.join(raw, d, "*", "*counter_collection.csv")))
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in files:
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: list(v.values()) for c, v in cs.items()} for k, cs in per.items()}


def mean(v):
    return sum(v) / len(v) if v else 0.0


def table(d, scale=1.0):
    res = {}
    for k, cs in collect(d).items():
        res[k] = {"dispatches": max(len(v) for v in cs.values())}
        for c, v in cs.items():
            res[k][c] = mean(v) * scale
    return res


# Issue cost of a wave64 VALU instruction on one SIMD, in cycles (tools/micro/valu_rate2.hip, in-kernel clock): full-rate
# arithmetic (fma / mul / add / sub, and / xor / ashr / add_u32 / mov) 2, half-rate (max / min / med3, cmp, cndmask, fract, perm, cvt,
# lshl / bfe / bfi, div_scale / fmas / fixup, readfirstlane) 4, transcendental (rcp / rsq / sqrt) 8.  The counters split the
# instructions into ADD / MUL / FMA / TRANS f32, INT32, INT64, CVT and a rest (total - these): compares, selects, min / max, moves,
# bit operations not counted as INT32.  INT32 and the rest are mixtures of 2- and 4-cycle instructions: priced at both.
PRICE_LO = {"ADD_F32": 2, "MUL_F32": 2, "FMA_F32": 2, "TRANS_F32": 8, "CVT": 4, "INT32": 2, "INT64": 4, "other": 2}
PRICE_HI = {"ADD_F32": 2, "MUL_F32": 2, "FMA_F32": 2, "TRANS_F32": 8, "CVT": 4, "INT32": 4, "INT64": 8, "other": 4}
CLASSES = ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT")


def class_bound(cls, cycles):
    """cls: {counter: mean per dispatch}; cycles: kernel cycles of the same kernel (another pass) -> classes, issue-limited
    cycles per SIMD at both pricings and as fractions of the kernel's cycles"""
    total = cls.get("SQ_INSTS_VALU", 0.0)
    c = {k: cls.get("SQ_INSTS_VALU_" + k, 0.0) for k in CLASSES}
    c["other"] = max(0.0, total - sum(c.values()))
    lo = sum(c[k] * PRICE_LO[k] for k in c) / SIMDS
    hi = sum(c[k] * PRICE_HI[k] for k in c) / SIMDS
    e = {"valu_by_class": {k: round(v) for k, v in c.items()}, "issue_cycles_per_simd_lo": lo, "issue_cycles_per_simd_hi": hi,
         "prices_lo": PRICE_LO, "prices_hi": PRICE_HI}
    if cycles:
        e["issue_frac_lo"] = lo / cycles
        e["issue_frac_hi"] = hi / cycles
    return e


summary = {"tag": tag, "note": "means per dispatch over the profiled run; WRITE_SIZE / FETCH_SIZE are KB in rocprofv3's output and are given in bytes here; "
                               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B); separate --pmc passes, no tracing"}
for key, wdir, fdir, sqdir, cdir in (("headline", "w", "f", "sq", "cls"), ("rgbf32", "f32w", "f32f", "f32sq", "f32cls"), ("config4", "c4w", "c4f", "c4sq", "c4cls"),
                                     ("shadow", "shw", "shf", "shsq", "shcls"), ("band8", "b8w", "b8f", "b8sq", "b8cls"),
                                     ("band8_overlapped", "b8ow", "b8of", "b8osq", "b8ocls")):
    w, f, sq, cl = table(wdir, 1024.0), table(fdir, 1024.0), table(sqdir), table(cdir)
    kernels = {}
    for k in sorted(set(w) | set(f) | set(sq)):
        if k.startswith("__amd") or "upload_kernel" in k or "Cijk" in k or k.startswith("at::") or "elementwise" in k:
            continue
        e = {}
        if k in w:
            e["write_bytes"] = w[k].get("WRITE_SIZE", 0.0)
        if k in f:
            e["fetch_bytes_raw"] = f[k].get("FETCH_SIZE", 0.0)
            e["fetch_bytes_corrected"] = 2.0 * e["fetch_bytes_raw"]
        if k in sq:
            s = sq[k]
            e["dispatches_profiled"] = s["dispatches"]
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
                if c in s:
                    e[c] = s[c]
            if "SQ_INSTS_VALU" in s and "GRBM_GUI_ACTIVE" in s:
                cyc = s["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
                e["kernel_cycles"] = cyc
                # issue-limited time at the full fp32 rate: one VALU wave-instruction per SIMD per 2 cycles (157 TFLOP/s spec)
                e["valu_issue_limited_cycles"] = s["SQ_INSTS_VALU"] / SIMDS * 2.0
                e["valu_issue_frac"] = e["valu_issue_limited_cycles"] / cyc
                # SQ_ACTIVE_INST_VALU counts quad-cycles of VALU activity over all SIMDs
                e["valu_busy_frac"] = s.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / SIMDS / cyc
                e["cycles_per_valu_inst"] = s.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / s["SQ_INSTS_VALU"]
        if k in cl and "SQ_INSTS_VALU" in cl[k]:
            e.update(class_bound(cl[k], e.get("kernel_cycles")))
        kernels[k] = e
    summary[key] = kernels
h = summary["headline"]
def call_totals(tab, frames, bpp):
    box = {k: v for k, v in tab.items() if k.startswith("box_")}
    return {
        "kernels": sorted(box),
        "write_bytes_per_call": sum(v.get("write_bytes", 0.0) for v in box.values()),
        "fetch_bytes_per_call_corrected": sum(v.get("fetch_bytes_corrected", 0.0) for v in box.values()),
        "valu_wave_insts_per_call": sum(v.get("SQ_INSTS_VALU", 0.0) for v in box.values()),
        "salu_wave_insts_per_call": sum(v.get("SQ_INSTS_SALU", 0.0) for v in box.values()),
        "valu_active_quad_cycles_per_call": sum(v.get("SQ_ACTIVE_INST_VALU", 0.0) for v in box.values()),
        "kernel_cycles_per_call": sum(v.get("kernel_cycles", 0.0) for v in box.values()),
        "valu_by_class": {c: sum(v.get("valu_by_class", {}).get(c, 0) for v in box.values()) for c in CLASSES + ("other",)},
        "issue_cycles_per_simd_lo": sum(v.get("issue_cycles_per_simd_lo", 0.0) for v in box.values()),
        "issue_cycles_per_simd_hi": sum(v.get("issue_cycles_per_simd_hi", 0.0) for v in box.values()),
        "frames_per_call": frames, "algorithmic_bytes_per_call": 1920 * 1080 * bpp * frames}


summary["rgbf32_call"] = call_totals(summary["rgbf32"], 160, 12)
# one rank's share of an 8-GPU step: rank 0's 136 of 1080 rows of the 160 frames
summary["band8_call"] = dict(call_totals(summary["band8"], 160, 4), algorithmic_bytes_per_call=1920 * 136 * 4 * 160, world=8, rows=136)
# ... in the launch shape of callers that overlap their calls (nt_render_opts.overlapped: 64-row waves), profiled one call at a time
summary["band8_overlapped_call"] = dict(call_totals(summary["band8_overlapped"], 160, 4), algorithmic_bytes_per_call=1920 * 136 * 4 * 160, world=8, rows=136)
summary["headline_call"] = call_totals(h, 160, 4)
c4 = {k: v for k, v in summary["config4"].items() if k.startswith("composite_packet")}
if c4:
    k, v = sorted(c4.items())[0]
    summary["config4_call"] = {"kernel": k, "frames_per_call": 8, **{a: b for a, b in v.items()}}
# kernel durations of the traced runs, per (kernel, grid): the bench mixes full-frame, one-eighth-band and fp32 calls
for d, name in (("kt", "bench"), ("ktall", "bench_all"), ("b8", "band8")):
    rows = collections.defaultdict(list)
    for f in newest(glob.glob(os.path.join(raw, d, "*", "*_kernel_trace.csv"))):
        for r in csv.DictReader(open(f)):
            rows[(short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))].append(
                (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    with open(os.path.join(out, "%s_%s_kernel_durations.csv" % (tag, name)), "w") as fh:
        fh.write("kernel,grid_x,grid_y,grid_z,calls,avg_us,min_us,max_us\n")
        for (k, gx, gy, gz), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            fh.write('"%s",%s,%s,%s,%d,%.2f,%.2f,%.2f\n' % (k, gx, gy, gz, len(v), sum(v) / len(v), min(v), max(v)))
            if name == "bench" and k.startswith("box_") and "<6, false" in k and gz.isdigit() and int(gz) >= 160 and len(v) >= 40:     # (160 frames + the lead slots)
                # (the full-frame calls of the timed loop: the largest grid of that kernel)
                cur = summary.setdefault("headline_kernel_us", {})
                cur[k] = max(cur.get(k, 0.0), round(sum(v) / len(v), 2))
json.dump(summary, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1)
with open(os.path.join(out, tag + "_sq_counters.txt"), "w") as fh:
    for key in ("headline", "rgbf32", "config4", "shadow", "band8", "band8_overlapped"):
        fh.write("== %s\n" % key)
        for k, v in summary[key].items():
            fh.write(k + "\n")
            for c, x in v.items():
                fh.write("    %-28s %s\n" % (c, ("%.4f" % x) if isinstance(x, float) and x < 100 else ("%.1f" % x if isinstance(x, float) else x)))
print(json.dumps(summary.get("headline_call"), indent=1))

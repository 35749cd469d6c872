#!/usr/bin/env python3
"""Where does the time of a call go around the tile kernel?  Reads a rocprofv3 --kernel-trace CSV of a bench run and prints, for
the dispatches of box_tile_kernel<6,false,64,1>, their durations and the gap from the end of the previous one to their start,
by what ran just before them (upload_kernel: a call with host cameras; another tile kernel: a call from a camera table).
    python3 tools/gap_probe.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
groups = {}
last_tile_end = None
for i, (name, s, e) in enumerate(ks):
    if name.startswith("box_tile_kernel<6, false, 64, 1>"):
        before = ks[i - 1][0][:24] if i else "-"
        if last_tile_end is not None:
            groups.setdefault(before, []).append(((s - last_tile_end) / 1e3, (e - s) / 1e3))
        last_tile_end = e
for k, v in groups.items():
    if len(v) < 5:
        continue
    gaps = sorted(x[0] for x in v)
    durs = sorted(x[1] for x in v)
    print("before it: %-26s n=%4d  previous tile end -> start: median %6.1f us (min %6.1f)   tile kernel: median %.1f us (min %.1f, max %.1f)" %
          (k, len(v), gaps[len(gaps) // 2], gaps[0], durs[len(durs) // 2], durs[0], durs[-1]))

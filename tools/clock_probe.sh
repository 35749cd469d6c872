#!/bin/bash
# Shader clock during the BoxScene kernels as a function of frames per call (is the "fixed" part of a call's time a clock
# effect?): GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration.   tools/clock_probe.sh "160 320"
fs=${1:-"160 320"}
root=$(pwd); out=$root/gpurun_out/clock_probe; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for f in $fs; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/f$f -- python3 $root/tools/band_proxy.py --world 1 --frames $f --steps 6 --warmup 2 > $out/f$f.log 2>&1
  python3 - $out/f$f $f <<'P'
import csv, glob, sys, collections
d, f = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    name, us = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], None))
    if us and "box_" in name:
        agg[name.split("(")[0][-40:]].append((float(r["Counter_Value"]) / 8.0, us))
for k, v in agg.items():
    cyc = sum(a for a, _ in v) / len(v); us = sum(b for _, b in v) / len(v)
    print("frames %s %-42s %9.0f cycles  %8.1f us  -> %.3f GHz" % (f, k, cyc, us, cyc / us / 1e3))
P
done

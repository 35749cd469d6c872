#!/bin/bash
# kernel durations of a BoxScene call as a function of the frames per call: tools/frames_prof.sh "80 160 320" [band_proxy args]
fs=$1; shift
root=$(pwd); out=$root/gpurun_out/frames_prof; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for f in $fs; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/f$f -- python3 $root/tools/band_proxy.py --world 1 --frames $f --steps 20 --warmup 3 "$@" > $out/f$f.log 2>&1
  echo "== frames $f: $(grep '^{' $out/f$f.log | sed 's/.*"event_us_per_call": \([0-9.]*\).*/events \1 us/')"
  cat $out/f$f/*/*_kernel_stats.csv | cut -d, -f1-4 | grep "box_\|upload" | sed 's/void (anonymous namespace):://; s/(NtCameraFixed.*)"/"/'
done

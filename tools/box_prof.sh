#!/bin/bash
# rocprofv3 kernel stats of the BoxScene paths: tools/box_prof.sh <outdir>
out=$(realpath ${1:-gpurun_out/box_prof})
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for p in 0 1; do
  for w in 1 8; do
    NTRACER_BOX_PATH=$p rocprofv3 --kernel-trace --stats --output-format csv -d $out/p${p}_w${w} -- python3 $root/tools/band_proxy.py --world $w > $out/p${p}_w${w}.log 2>&1
  done
  NTRACER_BOX_PATH=$p rocprofv3 --kernel-trace --stats --output-format csv -d $out/p${p}_f32 -- python3 $root/tools/band_proxy.py --world 1 --f32 --frames 32 --steps 20 > $out/p${p}_f32.log 2>&1
done
cd $out
for d in p*_w* p*_f32; do [ -d $d ] && { echo "== $d"; grep "^{" $d.log; cat $d/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | sed 's/void (anonymous namespace):://; s/(NtCameraFixed.*)"/"/' ; }; done

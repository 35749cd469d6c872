#!/bin/bash
# Counters and kernel times of two builds of the library on one GPU box: tools/lib_pmc_ab.sh <libA.so> <libB.so> [band_proxy args]
a=$1; b=$2; shift 2
root=$(pwd)
out=$root/gpurun_out/lib_pmc_ab
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for lib in $a $b; do
  export NTRACER_HIP_LIB=$root/$lib
  tag=$(basename $lib .so)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/$tag.pmc -- python3 $root/tools/band_proxy.py --steps 3 --warmup 1 "$@" > $out/$tag.pmc.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag.kt -- python3 $root/tools/band_proxy.py --steps 20 --warmup 3 "$@" > $out/$tag.kt.log 2>&1
  echo "== $tag"
  python3 $root/tools/pmc_sum.py $(ls $out/$tag.pmc/*/*counter_collection.csv | head -1) 4
  cat $out/$tag.kt/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | sed 's/void (anonymous namespace):://; s/(NtCameraFixed.*)"/"/'
done

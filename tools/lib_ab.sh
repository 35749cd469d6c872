#!/bin/bash
# A/B of two builds of the library on one GPU box: tools/lib_ab.sh <libA.so> <libB.so> [out]  (interleaved, twice each)
a=$1; b=$2; out=${3:-gpurun_out/lib_ab.log}
: > $out
for rep in 1 2; do
  for lib in $a $b; do
    export NTRACER_HIP_LIB=$PWD/$lib
    echo "# $lib" >> $out
    python3 tools/band_proxy.py --world 1 >> $out 2>&1
    python3 tools/band_proxy.py --world 8 >> $out 2>&1
    python3 tools/band_proxy.py --world 1 --f32 --frames 160 --steps 20 >> $out 2>&1
    python3 tools/band_proxy.py --world 1 --n 3 >> $out 2>&1
  done
done
grep "^{\|^#" $out | sed 's/"rank": 0, "band_rows": 8, //; s/"wall_us_per_call".*//; s/"box_path": "1", //'

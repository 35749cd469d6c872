#!/bin/bash
# A/B of the BoxScene paths on one GPU: NTRACER_BOX_PATH=0 (cull / box / redo kernels), 1 (fused tile kernel + redo kernel)

out=${1:-gpurun_out/box_ab.log}
paths=${2:-"0 1"}
: > $out
for p in $paths; do
  export NTRACER_BOX_PATH=$p
  echo "# path $p" >> $out
  python3 tools/band_proxy.py --world 1 >> $out 2>&1
  python3 tools/band_proxy.py --world 8 >> $out 2>&1
  python3 tools/band_proxy.py --world 1 --f32 --frames 32 --steps 20 >> $out 2>&1
  python3 tools/band_proxy.py --world 1 --n 3 >> $out 2>&1
done
grep "^{\|^#" $out | sed 's/"rank": 0, "band_rows": 8, //; s/"wall_us_per_call".*//'

#!/bin/bash
# SQ instruction counters of the BoxScene kernels: tools/box_pmc.sh <outdir> "<paths>" [band_proxy args]
out=$(realpath -m ${1:-gpurun_out/box_pmc})
paths=${2:-"0 1"}
shift 2
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for p in $paths; do
  NTRACER_BOX_PATH=$p rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p$p -- python3 $root/tools/band_proxy.py --steps 3 --warmup 1 "$@" > $out/p$p.log 2>&1
  echo "== path $p $@"
  python3 $root/tools/pmc_sum.py $(ls $out/p$p/*/*counter_collection.csv | head -1) 4
done

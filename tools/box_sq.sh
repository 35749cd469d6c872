#!/bin/bash
# SQ counters of the headline kernels in three passes: tools/box_sq.sh <outdir> [band_proxy args]
out=$(realpath -m ${1:-gpurun_out/box_sq}); shift
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $root/tools/band_proxy.py --steps 3 --warmup 1 "$@" > $out/p$i.log 2>&1
  echo "== pass $i"; tail -2 $out/p$i.log | cut -c1-200
  f=$(ls $out/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 $root/tools/pmc_sum.py $f 4
done

#!/bin/bash
# rocprofv3 kernel stats of the BoxScene kernels at the bench sizes: tools/box_prof2.sh <outdir> ["n list"]
out=$(realpath ${1:-gpurun_out/box_prof2})
ns=${2:-"6 3"}
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for n in $ns; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/n${n} -- python3 $root/tools/band_proxy.py --world 1 --n $n --steps 20 --warmup 3 > $out/n${n}.log 2>&1
done
cd $out
for n in $ns; do echo "== n $n"; grep "^{" n$n.log; cat n$n/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | sed 's/void (anonymous namespace):://; s/(NtCameraFixed.*)"/"/' ; done

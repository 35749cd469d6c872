set -x
tools/micro/build/d2h_paths > gpurun_out/r3_d2h.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r3_counters.txt 2>&1; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in ntracer_amd/libntracer_hip.so build_ab/occ7.so; do
  echo "# $lib" >> gpurun_out/r3_occ7_ab.log
  NTRACER_HIP_LIB=$PWD/$lib python tools/il_ab.py --values 1 --rounds 5 --cases head,f32,box3 >> gpurun_out/r3_occ7_ab.log 2>&1
done
done
python tools/il_ab.py --var NTRACER_BOX_LEAD --values 1,8,16,32 --frames 32 --cases head > gpurun_out/r3_lead_ab2.log 2>&1
python tools/il_ab.py --var NTRACER_BOX_LEAD --values 1,16,32,64 --frames 64 --cases head >> gpurun_out/r3_lead_ab2.log 2>&1
python tools/il_ab.py --var NTRACER_BOX_LEAD --values 1,16,32,64 --frames 320 --cases head >> gpurun_out/r3_lead_ab2.log 2>&1

#!/bin/bash
# Every rocprofv3 pass behind the numbers in bench.py / DESIGN.md, on one MI355X:   tools/profile_round.sh r02
# Writes raw output under gpurun_out/<tag>_prof/ and the summaries to commit under gpurun_out/<tag>_profiles/
# (copy those into profiles/).  Counters are collected in passes of their own (no tracing), per MI355X_MICROARCH.md.
tag=${1:-r03}
root=$(pwd)
raw=$root/gpurun_out/${tag}_prof
out=$root/gpurun_out/${tag}_profiles
mkdir -p $raw $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline"
S="python3 $root/bench.py --headline-only --steps 3 --warmup 1"
echo "[1] kernel trace of the headline loop (one stream, one kernel at a time: the timed region behind value and roofline)"
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/kt -- python3 $root/bench.py --headline-only > $raw/kt.log 2>&1
cp $(ls $raw/kt/*/*_kernel_stats.csv | head -1) $out/${tag}_bench_kernel_stats.csv
grep "^{" $raw/kt.log > $out/${tag}_bench_under_profiler.json
cp $root/gpurun_out/bench_details.json $out/${tag}_bench_under_profiler_details.json
echo "[1b] kernel trace of the whole bench (also the two-stream loops, whose kernels overlap and so last longer each; fp32, bands, config 5, extras)"
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/ktall -- $B > $raw/ktall.log 2>&1
cp $(ls $raw/ktall/*/*_kernel_stats.csv | head -1) $out/${tag}_bench_all_kernel_stats.csv
grep "^{" $raw/ktall.log > $out/${tag}_bench_all_under_profiler.json
cp $root/gpurun_out/bench_details.json $out/${tag}_bench_all_under_profiler_details.json
echo "[2] HBM bytes: WRITE_SIZE and FETCH_SIZE, separate passes"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $raw/w -- $S > $raw/w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $raw/f -- $S > $raw/f.log 2>&1
echo "[3] SQ counters of the headline kernels"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $raw/sq -- $S > $raw/sq.log 2>&1
echo "[3a] VALU instructions by class (ADD / MUL / FMA / TRANS f32, INT32, INT64, CVT; the rest = total - these)"
CLS="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
rocprofv3 --pmc $CLS --output-format csv -d $raw/cls -- $S > $raw/cls.log 2>&1
echo "[3b] the fp32x3 format of the same workload: SQ counters and bytes"
F32="python3 $root/tools/band_proxy.py --world 1 --f32 --frames 160 --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $raw/f32sq -- $F32 > $raw/f32sq.log 2>&1
rocprofv3 --pmc $CLS --output-format csv -d $raw/f32cls -- $F32 > $raw/f32cls.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $raw/f32w -- $F32 > $raw/f32w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $raw/f32f -- $F32 > $raw/f32f.log 2>&1
echo "[4] config 4 (120-cell, composite_packet): SQ counters, then bytes"
C4="python3 $root/tools/run_composite.py cell120_n4 8"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $raw/c4sq -- $C4 > $raw/c4sq.log 2>&1
rocprofv3 --pmc $CLS --output-format csv -d $raw/c4cls -- $C4 > $raw/c4cls.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $raw/c4w -- $C4 > $raw/c4w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $raw/c4f -- $C4 > $raw/c4f.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/c4kt -- $C4 > $raw/c4kt.log 2>&1
cp $(ls $raw/c4kt/*/*_kernel_stats.csv | head -1) $out/${tag}_config4_kernel_stats.csv
echo "[5] the one-eighth band of the headline workload (what a rank of an 8-GPU run does)"
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/b8 -- python3 $root/tools/band_proxy.py --world 8 > $raw/b8.log 2>&1
cp $(ls $raw/b8/*/*_kernel_stats.csv | head -1) $out/${tag}_band8_kernel_stats.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $raw/b8w -- python3 $root/tools/band_proxy.py --world 8 --steps 5 --warmup 2 > $raw/b8w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $raw/b8f -- python3 $root/tools/band_proxy.py --world 8 --steps 5 --warmup 2 > $raw/b8f.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $raw/b8sq -- python3 $root/tools/band_proxy.py --world 8 --steps 5 --warmup 2 > $raw/b8sq.log 2>&1
rocprofv3 --pmc $CLS --output-format csv -d $raw/b8cls -- python3 $root/tools/band_proxy.py --world 8 --steps 5 --warmup 2 > $raw/b8cls.log 2>&1
echo "[5b] the same with nt_render_opts.overlapped (the launch shape of bench.py's two-stream legs: 64-row waves), one call at a time under the profiler"
B8O="python3 $root/tools/band_proxy.py --world 8 --overlapped --steps 5 --warmup 2"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $raw/b8ow -- $B8O > $raw/b8ow.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $raw/b8of -- $B8O > $raw/b8of.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $raw/b8osq -- $B8O > $raw/b8osq.log 2>&1
rocprofv3 --pmc $CLS --output-format csv -d $raw/b8ocls -- $B8O > $raw/b8ocls.log 2>&1
echo "[6] the 120-cell with lights and shadows at 1920x1080 (packet pass + shading pass): trace, SQ counters, classes, bytes"
SH="python3 $root/tools/run_shadow.py 1920 1080 8"
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/shkt -- $SH > $raw/shkt.log 2>&1
cp $(ls $raw/shkt/*/*_kernel_stats.csv | head -1) $out/${tag}_shadow_kernel_stats.csv
grep "shadow scene" $raw/shkt.log | tail -1 > $out/${tag}_shadow_under_profiler.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $raw/shsq -- $SH > $raw/shsq.log 2>&1
rocprofv3 --pmc $CLS --output-format csv -d $raw/shcls -- $SH > $raw/shcls.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $raw/shwait -- $SH > $raw/shwait.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $raw/shw -- $SH > $raw/shw.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $raw/shf -- $SH > $raw/shf.log 2>&1
echo "[7] issue cost of the instruction classes, with the clock the chip held (tools/micro/valu_rate2.hip)"
$root/tools/micro/build/valu_rate2 > $out/${tag}_valu_rate2.txt 2>&1
cd $root
python3 tools/pmc_summary.py $raw $out $tag
ls -la $out

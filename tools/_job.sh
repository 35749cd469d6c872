python -m pytest tests -m gpu -x -q > gpurun_out/r3_t5.log 2>&1; tail -4 gpurun_out/r3_t5.log
for n in 16 17 20 24; do python tools/boxn_time.py $n 64 5 2>&1 | grep Box; done
for n in 17 24; do NTRACER_FORCE_VAR=1 python tools/boxn_time.py $n 64 3 2>&1 | grep Box; done

mkdir -p gpurun_out
L=$PWD/astar-pairwise-aligner_amd/libastarpa_c_hip_timers.so
T=${1:-x}
for d in 1; do
for e in 0.01 0.15; do echo "== depth $d e $e"; PA_LIB_PATH=$L PA_SWEEP_TIMING=1 PA_SWEEP_DEPTH=$d python tests/tools/dropin_phases.py 10000 $e 2>&1 | tail -6; done
done > gpurun_out/r06_timers_10k_$T.log 2>&1
PA_LIB_PATH=$L PA_SWEEP_TIMING=1 PA_SWEEP_DEPTH=1 python tools/c3_quick.py 2>&1 | tail -8 > gpurun_out/r06_timers_c3_d1_$T.log
for e in 0.01 0.05 0.10 0.15; do python tests/tools/dropin_phases.py 10000 $e 2>&1 | tail -1; done > gpurun_out/r06_dropin_$T.log
python tools/c3_quick.py >> gpurun_out/r06_dropin_$T.log 2>&1
python -m pytest tests/test_gpu_sweep.py -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/r06_dropin_$T.log
cut -c1-250 gpurun_out/r06_timers_10k_$T.log; cut -c1-250 gpurun_out/r06_timers_c3_d1_$T.log; cat gpurun_out/r06_dropin_$T.log

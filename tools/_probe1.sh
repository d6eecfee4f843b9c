set -x
mkdir -p gpurun_out
PA_ALIGN_PROFILE=1 python tools/strong_probe.py 100000 > gpurun_out/r06_strong_probe.log 2>&1
for e in 0.01 0.05 0.10 0.15; do PA_SWEEP_TIMING=1 python tests/tools/dropin_phases.py 10000 $e; done > gpurun_out/r06_dropin_phases.log 2>&1
tail -50 gpurun_out/r06_strong_probe.log

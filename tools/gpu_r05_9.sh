cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 120 python tools/combine_probe.py 16 6 > gpurun_out/r05/combine_probe.log 2>&1
timeout 120 python tools/combine_probe.py 64 4 >> gpurun_out/r05/combine_probe.log 2>&1
PA_ALIGN_PROFILE=1 timeout 120 python tools/combine_probe.py 16 3 >> gpurun_out/r05/combine_probe.log 2>&1
cat gpurun_out/r05/combine_probe.log | cut -c1-260

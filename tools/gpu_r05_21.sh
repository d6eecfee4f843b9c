cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
PA_APA2_RDV=2 timeout 400 python tests/tools/fuzz_apa2.py 150 20260930 > gpurun_out/r05/fuzz_apa2_rdv.log 2>&1; echo "rc=$?" >> gpurun_out/r05/fuzz_apa2_rdv.log; tail -4 gpurun_out/r05/fuzz_apa2_rdv.log
PA_APA2_RDV=2 PA_APA2_RDV_PATIENCE_US=200 timeout 400 python tests/tools/fuzz_restated_gpu.py 120 20260931 16 > gpurun_out/r05/fuzz_restated_rdv.log 2>&1; echo "rc=$?" >> gpurun_out/r05/fuzz_restated_rdv.log; tail -4 gpurun_out/r05/fuzz_restated_rdv.log

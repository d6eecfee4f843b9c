set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_apa2_batch.py tests/test_gpu_apa2_full.py -x -q > gpurun_out/r05/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/t1.log; tail -5 gpurun_out/r05/t1.log
for rdv in 0 1; do
  PA_APA2_RDV=$rdv timeout 300 python tools/apa2_bench.py simple 10000 4096 > gpurun_out/r05/bench_simple_rdv$rdv.log 2>&1
  PA_APA2_RDV=$rdv timeout 300 python tools/apa2_bench.py full 10000 4096 > gpurun_out/r05/bench_full_rdv$rdv.log 2>&1
done
for us in 5 50; do
  PA_APA2_RDV_PATIENCE_US=$us timeout 300 python tools/apa2_bench.py simple 10000 > gpurun_out/r05/bench_simple_p$us.log 2>&1
  PA_APA2_RDV_PATIENCE_US=$us timeout 300 python tools/apa2_bench.py full 10000 4096 > gpurun_out/r05/bench_full_p$us.log 2>&1
done
grep -h "pairs/s\|half-wave" gpurun_out/r05/bench_*.log

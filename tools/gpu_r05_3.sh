cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_restated_fixtures.py -x -q --durations=8 > gpurun_out/r05/t_fixtures.log 2>&1; echo "rc=$?" >> gpurun_out/r05/t_fixtures.log; tail -15 gpurun_out/r05/t_fixtures.log
timeout 600 python -m pytest tests/test_gpu_apa2_batch.py tests/test_gpu_apa2_full.py tests/test_gpu_batch_align.py -x -q > gpurun_out/r05/t3.log 2>&1; echo "rc=$?" >> gpurun_out/r05/t3.log; tail -4 gpurun_out/r05/t3.log
L=gpurun_out/r05/order_probe2.log
: > $L
for preset in simple full; do
  for rdv in 0 1; do
    PA_APA2_RDV=$rdv timeout 200 python tools/order_probe.py $preset 10000 mixed >> $L 2>&1
  done
  PA_APA2_ORDER_LENGTH=1 timeout 200 python tools/order_probe.py $preset 10000 mixed >> $L 2>&1
done
cat $L
timeout 300 python tools/apa2_bench.py simple 10000 512 4096 > gpurun_out/r05/bench_simple_sketch.log 2>&1
timeout 300 python tools/apa2_bench.py full 10000 512 4096 > gpurun_out/r05/bench_full_sketch.log 2>&1
grep -h "pairs/s\|half-wave\|created again" gpurun_out/r05/bench_*_sketch.log

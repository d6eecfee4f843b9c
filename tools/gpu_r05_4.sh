cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/prio_probe.log
: > $L
for preset in simple full; do
  for prio in 0 1; do
    PA_APA2_PRIO=$prio timeout 200 python tools/order_probe.py $preset 10000 mixed >> $L 2>&1
  done
done
PA_APA2_PRIO=0 timeout 300 python tools/apa2_bench.py full 1000 4096 2>&1 | grep "pairs/s" >> $L
PA_APA2_PRIO=1 timeout 300 python tools/apa2_bench.py full 1000 4096 2>&1 | grep "pairs/s" >> $L
PA_APA2_PRIO=0 timeout 300 python tools/apa2_bench.py simple 1000 4096 2>&1 | grep "pairs/s" >> $L
PA_APA2_PRIO=1 timeout 300 python tools/apa2_bench.py simple 1000 4096 2>&1 | grep "pairs/s" >> $L
cat $L
timeout 900 python -m pytest tests/test_gpu_restated_fixtures.py -x -q --durations=5 > gpurun_out/r05/t_fixtures.log 2>&1; echo "rc=$?" >> gpurun_out/r05/t_fixtures.log; tail -12 gpurun_out/r05/t_fixtures.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_restated_fixtures.py > gpurun_out/r05/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/r05/t_all.log; tail -8 gpurun_out/r05/t_all.log

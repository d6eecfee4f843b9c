cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
PA_APA2_PROBE_STATS=1 timeout 300 python tools/apa2_bench.py full 10000 4096 > gpurun_out/r05/pair_times.log 2>&1
grep "apa2_full\]" gpurun_out/r05/pair_times.log | tail -20
echo "== drop-in threads (at most 8 callers on the single-pair path)" > gpurun_out/r05/dropin_threads2.log
timeout 300 python tools/dropin_threads.py --pairs 1280 --threads 1,8,16,32,64 >> gpurun_out/r05/dropin_threads2.log 2>&1
echo "== PA_SINGLE_MAX=1" >> gpurun_out/r05/dropin_threads2.log
PA_SINGLE_MAX=1 timeout 300 python tools/dropin_threads.py --pairs 1280 --threads 8,16,64 >> gpurun_out/r05/dropin_threads2.log 2>&1
echo "== PA_SINGLE_MAX=4" >> gpurun_out/r05/dropin_threads2.log
PA_SINGLE_MAX=4 timeout 300 python tools/dropin_threads.py --pairs 1280 --threads 8,16,64 >> gpurun_out/r05/dropin_threads2.log 2>&1
cat gpurun_out/r05/dropin_threads2.log
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q 2>&1 | tail -3

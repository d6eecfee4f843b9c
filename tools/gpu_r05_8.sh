cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
echo "== drop-in threads (crowded mode from 12 callers, sticky 20 ms)" > gpurun_out/r05/dropin_threads3.log
timeout 300 python tools/dropin_threads.py --pairs 1280 --threads 1,8,16,32,64 >> gpurun_out/r05/dropin_threads3.log 2>&1
echo "== PA_COMBINE_MIN=9" >> gpurun_out/r05/dropin_threads3.log
PA_COMBINE_MIN=9 timeout 300 python tools/dropin_threads.py --pairs 1280 --threads 16,32,64 >> gpurun_out/r05/dropin_threads3.log 2>&1
cat gpurun_out/r05/dropin_threads3.log
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q 2>&1 | tail -3

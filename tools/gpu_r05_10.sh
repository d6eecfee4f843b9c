cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
echo "== drop-in threads (crowded mode from 12 callers; gather window 300 us, up to 8 batches side by side; pooled batch stream)" > gpurun_out/r05/dropin_threads4.log
timeout 300 python tools/dropin_threads.py --pairs 1280 --threads 1,8,16,32,64 >> gpurun_out/r05/dropin_threads4.log 2>&1
cat gpurun_out/r05/dropin_threads4.log
timeout 120 python tools/combine_probe.py 16 4 2>&1 | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_batch_align.py -x -q 2>&1 | tail -3

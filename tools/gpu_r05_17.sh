cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
PA_POISON_ALLOC=1 timeout 1200 python -m pytest tests/test_gpu_apa2_batch.py tests/test_gpu_apa2_full.py tests/test_gpu_batch_align.py tests/test_gpu_engine.py tests/test_gpu_operator.py tests/test_gpu_restated_fixtures.py -x -q > gpurun_out/r05/t_poison.log 2>&1; echo "rc=$?" >> gpurun_out/r05/t_poison.log; tail -5 gpurun_out/r05/t_poison.log

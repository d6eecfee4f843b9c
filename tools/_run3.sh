cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02d
timeout 900 python -m pytest tests/test_gpu_batch_align.py tests/test_gpu_fullsize.py -m gpu -x -q --durations=5 > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r02d/pytest.log
timeout 600 python tools/shape_sweep.py 41 146 256 384 512 768 1024 1536 2048 > gpurun_out/r02d/shape_sweep.log 2>&1
cat gpurun_out/r02d/shape_sweep.log

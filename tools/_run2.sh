cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests/test_gpu_operator.py tests/test_gpu_batch_align.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r02c/pytest.log
CASES="100000x1 100000x8 100000x16 100000x41 100000x64 100000x146 100000x256 100000x512 100000x1024 100000x2048 100000x3072 100000x4096 100000x5120 100000x6144 10000x10000"
python tools/quick_bench.py $CASES > gpurun_out/r02c/quick_ldseq.log 2>&1
PA_STRIP_NO_LDSEQ=1 PA_PAIR_NO_LDSEQ=1 python tools/quick_bench.py $CASES > gpurun_out/r02c/quick_noldseq.log 2>&1
paste -d'\n' gpurun_out/r02c/quick_ldseq.log gpurun_out/r02c/quick_noldseq.log | cut -c1-150

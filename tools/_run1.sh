cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r02b/pytest.log
FAST="--no-cpu-baseline --no-single-pair --no-c4 --no-banded --no-engine --no-c5 --no-c4-sharded"
python bench.py --steps 3 --warmup 1 $FAST > gpurun_out/r02b/bench_ldseq.json 2> gpurun_out/r02b/bench_ldseq.err; tail -1 gpurun_out/r02b/bench_ldseq.json | cut -c1-600
PA_PAIR_NO_LDSEQ=1 python bench.py --steps 3 --warmup 1 $FAST > gpurun_out/r02b/bench_noldseq.json 2> gpurun_out/r02b/bench_noldseq.err; tail -1 gpurun_out/r02b/bench_noldseq.json | cut -c1-600

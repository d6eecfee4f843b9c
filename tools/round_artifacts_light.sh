#!/bin/bash
# The end-of-round pair of artifacts, without the PMC passes (tools/round_artifacts.sh has those): the bench line and the kernel trace of
# the same command.   bash tools/round_artifacts_light.sh <tag>  -> gpurun_out/<tag>/
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o bench -- python bench.py --no-cpu-baseline --no-c5 --no-engine > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_rocprof.err
cp $OUT/bench_trace/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/bench_trace
ls -la $OUT

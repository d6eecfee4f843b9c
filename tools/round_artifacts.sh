#!/bin/bash
# Everything a round's profiles/ entry is made from, in one go on the GPU box: PMC passes of the headline kernel (tools/pmc_run.sh), of the
# batched A*PA2 kernels (tools/pmc_apa2.sh, both presets), the bench line, and the kernel trace of the bench command.
#   bash tools/round_artifacts.sh <tag>      -> gpurun_out/<tag>/
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/pmc_run.sh > $OUT/pmc_run.log 2>&1
cp gpurun_out/pmc/summary.json $OUT/pmc_headline_summary.json
cp gpurun_out/pmc/trace/*kernel_stats.csv $OUT/pmc_headline_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc/engine_trace/*kernel_stats.csv $OUT/engine_kernel_stats.csv 2>/dev/null
PA_APA2_PRESET=full bash tools/pmc_apa2.sh > $OUT/pmc_apa2_full.log 2>&1
cp gpurun_out/pmc_apa2_full/summary.json $OUT/pmc_apa2_full_summary.json
PA_APA2_PRESET=simple bash tools/pmc_apa2.sh > $OUT/pmc_apa2_simple.log 2>&1
cp gpurun_out/pmc_apa2_simple/summary.json $OUT/pmc_apa2_simple_summary.json
# the bench line with the fresh counters in place (bench.py reads profiles/pmc_latest.json)
cp gpurun_out/pmc/summary.json profiles/pmc_latest.json
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o bench -- python bench.py --no-cpu-baseline > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_rocprof.err
cp $OUT/bench_trace/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/bench_trace gpurun_out/pmc/*/ gpurun_out/pmc_apa2_*/*/ 2>/dev/null
ls -la $OUT

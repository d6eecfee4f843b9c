#!/bin/bash
# The headline kernel's counters in four short passes (kernel trace, FETCH_SIZE, WRITE_SIZE, the SQ instruction counters): what
# bench.py's roofline.traffic needs (profiles/pmc_latest.json, stamped with the kernel sources' hash).  tools/pmc_run.sh has all six
# passes and the engine legs.   bash tools/pmc_run_light.sh  -> gpurun_out/pmc/summary.json
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-pair --no-c4 --no-banded --no-engine --no-c5 --no-c4-sharded $PA_BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
python tools/pmc_summary.py $OUT > $OUT/summary_print.log 2>&1
rm -rf $OUT/trace/*/*.db $OUT/*/*/*agent_info.csv 2>/dev/null
ls -la $OUT

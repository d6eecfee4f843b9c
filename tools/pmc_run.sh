#!/bin/bash
# PMC passes for the bench workload (counters in their own runs: --kernel-trace only, no sys/hip traces).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-pair --no-c4 --no-banded --no-engine --no-c5 --no-c4-sharded $PA_BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/lds -o lds -- $CMD > $OUT/lds.log 2>&1
python tools/pmc_summary.py $OUT
# the engine legs (C3 presets, drop-in loop, C5 sweep) under the kernel trace: per-kernel time of sweep_kernel & co
ENG="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single-pair --no-c4 --no-banded --no-c4-sharded"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/engine_trace -o engine -- $ENG > $OUT/engine_trace.log 2>&1

#!/bin/bash
# rocprofv3 passes for the batched A*PA2 path (pa::apa2::apa2_kernel + pa::trace_kernel): kernel trace with statistics, then the
# counters in runs of their own (--kernel-trace only).  Workload: tools/apa2_bench.py <preset> <c4 pairs> <100 kbp pairs>; PA_APA2_PRESET=full for
# pa::apa2::apa2_full_kernel.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PRESET=${PA_APA2_PRESET:-simple}
OUT=gpurun_out/pmc_apa2_$PRESET
mkdir -p $OUT
CMD="python tools/apa2_bench.py $PRESET ${PA_APA2_C4:-10000} ${PA_APA2_C3:-4096}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
python tools/pmc_apa2_summary.py $OUT

#!/bin/bash
# Round 6: the same bit-sliced batch through two builds of the library under rocprofv3 counters (why is one slower?)
#   bash tools/slice_ab_counters.sh <libA.so> <libB.so>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tag in A B; do
  lib=$1; shift
  out=gpurun_out/ab_$tag
  rm -rf $out && mkdir -p $out
  PA_LIB_PATH=$lib PA_SLICE=50 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o r -- python tools/quick_bench.py 100000x8192 > $out/log.txt 2>&1
  tail -1 $out/log.txt
  python - $out <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'slice_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k:'%.4g'%(sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
done

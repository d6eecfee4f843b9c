#!/bin/bash
# HIP API + kernel statistics of the drop-in loop under 1 and 4 caller threads (rocprofv3; summaries into gpurun_out/dropin_prof).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/dropin_prof
for T in 1 4; do
  timeout 60 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/dp_t$T -o p -- python $R/tools/dropin_threads.py --pairs 100 --threads $T 2>&1 | grep "^threads" > $R/gpurun_out/dropin_prof/t$T.log
  for f in $(find /tmp/dp_t$T -name '*stats.csv'); do cp $f $R/gpurun_out/dropin_prof/t${T}_$(basename $f); done
done
cd $R/gpurun_out/dropin_prof && for f in t1.log t4.log t1_p_hip_api_stats.csv t4_p_hip_api_stats.csv t1_p_kernel_stats.csv t4_p_kernel_stats.csv; do echo "== $f"; [ -f $f ] && head -12 $f; done; true

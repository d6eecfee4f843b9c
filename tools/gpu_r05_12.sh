cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
gcc -O2 tests/c_abi/dropin_threads.c -Iinclude -Lastar-pairwise-aligner_amd -lastarpa_c_hip -lpthread -o /tmp/dropin_threads || exit 1
export LD_LIBRARY_PATH=$PWD/astar-pairwise-aligner_amd:$LD_LIBRARY_PATH GPU_MAX_HW_QUEUES=16
L=gpurun_out/r05/dropin_threads_c2.log
echo "== tests/c_abi/dropin_threads.c: astarpa2_simple, 1280 x 10 kbp pairs (pinned size classes)" > $L
timeout 300 /tmp/dropin_threads 1280 simple 16 32 64 128 >> $L 2>&1; echo "rc=$?" >> $L
cat $L
PA_ALIGN_PROFILE=1 timeout 300 /tmp/dropin_threads 640 simple 64 > gpurun_out/r05/dropin_profile.log 2>&1
grep -c "pa_batch_create\] host layout" gpurun_out/r05/dropin_profile.log
python - <<'PY'
import re,collections
agg=collections.defaultdict(list)
for ln in open('gpurun_out/r05/dropin_profile.log',errors='ignore'):
    m=re.match(r'\[(pa_batch_\w+)\]\s+(.*?)\s+([\d.]+) ms',ln)
    if m: agg[(m.group(1),m.group(2).strip())].append(float(m.group(3)))
for k,v in agg.items():
    v2=sorted(v); print(k, 'n',len(v),'median %.3f'%v2[len(v2)//2],'p90 %.3f'%v2[int(len(v2)*0.9)],'max %.3f'%v2[-1])
PY

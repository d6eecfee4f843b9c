cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
gcc -O2 tests/c_abi/dropin_threads.c -Iinclude -Lastar-pairwise-aligner_amd -lastarpa_c_hip -lpthread -o /tmp/dropin_threads || exit 1
export LD_LIBRARY_PATH=$PWD/astar-pairwise-aligner_amd:$LD_LIBRARY_PATH GPU_MAX_HW_QUEUES=16
L=gpurun_out/r05/dropin_threads_c.log
echo "== tests/c_abi/dropin_threads.c: astarpa2_simple, 1280 x 10 kbp pairs" > $L
timeout 300 /tmp/dropin_threads 1280 simple 8 16 32 64 128 >> $L 2>&1; echo "rc=$?" >> $L
echo "== PA_COMBINE=0" >> $L
PA_COMBINE=0 timeout 300 /tmp/dropin_threads 640 simple 8 16 32 >> $L 2>&1; echo "rc=$?" >> $L
echo "== astarpa2_full" >> $L
timeout 300 /tmp/dropin_threads 640 full 8 16 64 >> $L 2>&1; echo "rc=$?" >> $L
cat $L

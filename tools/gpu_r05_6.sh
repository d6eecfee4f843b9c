cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/phase_clocks.log
: > $L
for rdv in 0 1; do
  echo "== PA_APA2_RDV=$rdv full 4096 x 100 kbp, phase clocks" >> $L
  PA_APA2_RDV=$rdv PA_APA2_PROBE_STATS=1 timeout 300 python tools/apa2_bench.py full 10000 4096 2>&1 | grep -v "created again" >> $L
done
cat $L
echo "== drop-in threads" > gpurun_out/r05/dropin_threads.log
timeout 300 python tools/dropin_threads.py --pairs 1280 --threads 1,8,16,32,64 >> gpurun_out/r05/dropin_threads.log 2>&1
PA_COMBINE=0 timeout 300 python tools/dropin_threads.py --pairs 1280 --threads 16,64 >> gpurun_out/r05/dropin_threads.log 2>&1
cat gpurun_out/r05/dropin_threads.log
PA_ALIGN_PROFILE=1 timeout 200 python tools/apa2_bench.py simple 10000 2>&1 | grep "start order" | tail -3
PA_APA2_PRESET=full PA_APA2_C4=10000 PA_APA2_C3=4096 timeout 900 bash tools/pmc_apa2.sh > gpurun_out/r05/pmc_full.log 2>&1
tail -c 3000 gpurun_out/pmc_apa2_full/summary.json

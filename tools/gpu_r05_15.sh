cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_apa2_full.py tests/test_gpu_restated_fixtures.py -x -q 2>&1 | tail -4
L=gpurun_out/r05/search_windows.log
: > $L
for w in 0 1; do
  echo "== PA_APA2_SEARCH_WINDOWS=$w" >> $L
  PA_APA2_SEARCH_WINDOWS=$w PA_APA2_PROBE_STATS=1 timeout 300 python tools/apa2_bench.py full 10000 512 4096 2>&1 | grep -v "created again\|XCD" >> $L
  PA_APA2_SEARCH_WINDOWS=$w timeout 300 python tools/apa2_bench.py full 10000 512 4096 2>&1 | grep "pairs/s" >> $L
done
cut -c1-420 $L

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05/t_all2.log 2>&1; echo "rc=$?" >> gpurun_out/r05/t_all2.log; tail -6 gpurun_out/r05/t_all2.log
L=gpurun_out/r05/chunks.log
: > $L
for ch in 1 2 3; do
  echo "== PA_ALIGN_CHUNKS=$ch" >> $L
  PA_ALIGN_CHUNKS=$ch timeout 200 python tools/apa2_bench.py simple 10000 2>&1 | grep "pairs/s" >> $L
  PA_ALIGN_CHUNKS=$ch timeout 200 python tools/apa2_bench.py full 10000 2>&1 | grep "pairs/s" >> $L
done
cut -c1-200 $L

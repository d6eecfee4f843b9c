cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
L=gpurun_out/r05/waves5.log
echo "apa2_full_kernel built with -DPA_APA2_FULL_WAVES=5 (96 VGPRs, 2 spilled)" > $L
for per_cu in 4 5; do
  echo "== PA_APA2_BLOCKS_PER_CU=$per_cu" >> $L
  PA_APA2_BLOCKS_PER_CU=$per_cu timeout 300 python tools/apa2_bench.py full 10000 512 4096 2>&1 | grep "pairs/s" | cut -c1-170 >> $L
  PA_APA2_BLOCKS_PER_CU=$per_cu timeout 300 python tools/order_probe.py full 40000 mixed 2>&1 | tail -1 >> $L
done
cat $L

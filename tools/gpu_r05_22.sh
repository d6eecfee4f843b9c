cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
L=gpurun_out/r05/prio_probe2.log
: > $L
for preset in simple full; do
  for prio in 0 1 0 1; do
    echo "PA_APA2_PRIO=$prio" >> $L
    PA_APA2_PRIO=$prio timeout 200 python tools/order_probe.py $preset 10000 mixed >> $L 2>&1
  done
done
cat $L

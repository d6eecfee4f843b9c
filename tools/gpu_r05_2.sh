cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/order_probe.log
: > $L
for preset in simple full; do
  for n in 10000 40000; do
    for rdv in 0 1; do
      PA_APA2_RDV=$rdv timeout 200 python tools/order_probe.py $preset $n mixed >> $L 2>&1
      PA_APA2_RDV=$rdv PA_APA2_ORDER_INPUT=1 timeout 200 python tools/order_probe.py $preset $n sorted >> $L 2>&1
    done
  done
done
cat $L

cd $GRAFT_REPO_ROOT
PA_ALIGN_PROFILE=1 timeout 200 python tools/apa2_bench.py simple 10000 2>&1 | grep "pa_batch_create\]\|created again" | tail -24
PA_ALIGN_PROFILE=1 timeout 200 python tools/apa2_bench.py full 10000 2>&1 | grep "pa_batch_create\]\|created again" | tail -40

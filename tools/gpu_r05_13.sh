cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_apa2_full.py -x -q 2>&1 | tail -5
timeout 300 python tools/apa2_bench.py full 10000 4096 2>&1 | grep "pairs/s\|match building" | cut -c1-330 | tee gpurun_out/r05/build_chain.log
timeout 300 python tools/gcsh_build_clocks.py 2>&1 | tail -12 | tee -a gpurun_out/r05/build_chain.log

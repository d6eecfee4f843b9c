cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05b
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05b/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/r05b/t_all.log; tail -4 gpurun_out/r05b/t_all.log
PA_APA2_PRESET=full bash tools/pmc_apa2.sh > gpurun_out/r05b/pmc_apa2_full.log 2>&1
cp gpurun_out/pmc_apa2_full/summary.json gpurun_out/r05b/pmc_apa2_full_summary.json
bash tools/round_artifacts_light.sh r05b

cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05c
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05c/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/r05c/t_all.log; tail -4 gpurun_out/r05c/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/round_artifacts_light.sh r05c

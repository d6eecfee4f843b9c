cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/r05/t_all.log; tail -15 gpurun_out/r05/t_all.log
timeout 900 python bench.py > gpurun_out/r05/bench_a.json 2> gpurun_out/r05/bench_a.err; echo "bench rc=$?"; tail -3 gpurun_out/r05/bench_a.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_a.json'))
for k in ('value','ms_per_step','roofline','dropin_loop','c3_engine','c5','regressions'):
    print(k, json.dumps(d.get(k))[:600])
for k in d:
    if k.startswith('c4_astar') or k.startswith('c3_batch'):
        v=d[k]; print(k, {x:v.get(x) for x in ('pairs_per_sec','c_abi_pairs_per_sec','forward_kernel_ms','trace_kernel_ms','create_ms','gpu_match_building_ms','pairs_per_sec_incl_create_again','half_wave_blocks')})
PY

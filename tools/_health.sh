mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06c_gpu_suite.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r06c_gpu_suite.log
( time python bench.py > gpurun_out/r06c_bench_line.json 2> gpurun_out/r06c_bench.err ) 2>&1 | tail -4
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06c_bench_line.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline'].get('frac'))
print(json.dumps(j.get('c4_sharded'))[:1500])
print(json.dumps(j.get('dropin_loop'))[:800])
print(json.dumps(j.get('c3_engine'))[:800])
print(j.get('regressions'))
PY

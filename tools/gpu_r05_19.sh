cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-single-pair --no-c4 --no-banded --no-engine --no-c5 --no-apa2 --steps 1 --warmup 0 --pairs 64"
for i in 1 2; do python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['c4_sharded']['pairs_per_sec'], d['c4_sharded']['ms'], d['c4_sharded']['astarpa2_simple'])"; done
PA_ALIGN_PROFILE=1 python bench.py $F 2>&1 >/dev/null | grep -v "^\[pa_batch_destroy\] busy" | tail -60 | cut -c1-120

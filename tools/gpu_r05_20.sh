cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-single-pair --no-c4 --no-banded --no-engine --no-c5 --no-apa2 --steps 2 --warmup 1"
PA_ALIGN_PROFILE=1 python bench.py $F 2> gpurun_out/sharded_profile.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['c4_sharded']['pairs_per_sec'], d['c4_sharded']['ms'], d['c4_sharded']['astarpa2_simple'])"
grep -v "busy:" gpurun_out/sharded_profile.err | tail -75 | cut -c1-110

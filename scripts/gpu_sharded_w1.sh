# world-1 sharded legs: direct, machinery, machinery + unique rows
for w in inbatch triplet glove; do
  for mode in "direct:" "machinery:ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0" "unique:ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=1"; do
    name=${mode%%:*}; envs=${mode#*:}
    echo "$w $name: $(env ESR_BENCH_SHARDED=1 $envs python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -2

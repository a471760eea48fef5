# world-1 sharded legs: direct, machinery (per-occurrence plans), machinery + unique rows; one-call exchange halves on / off
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_rccl_world2.py -x -q 2>&1 | tail -2
mkdir -p gpurun_out/r4s
for w in inbatch triplet glove; do
  for mode in "direct:" "machinery:ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0" "unique:ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=1" "unique_opbyop:ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=1 ESR_SHARDED_FUSED=0"; do
    name=${mode%%:*}; envs=${mode#*:}
    env ESR_BENCH_SHARDED=1 $envs python bench.py --workload $w --steps 200 --warmup 24 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r4s/bench_sharded_world1_${name}_$w.json
    echo "$w $name: $(python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(d["ms_per_step"], d["value"])' gpurun_out/r4s/bench_sharded_world1_${name}_$w.json)"
  done
done

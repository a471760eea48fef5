# SURVEY 8e overlap (next batch's lookup under this batch's kernels + stale-row patch): GPU tests, then the world-1
# machinery legs with it off / on (world 1 has no link traffic to hide: these figures are the COST of the scheme)
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q 2>&1 | tail -3
mkdir -p gpurun_out/r4o
for w in triplet glove inbatch; do
  for mode in "off:ESR_SHARDED_OVERLAP=0" "on:ESR_SHARDED_OVERLAP=1"; do
    name=${mode%%:*}; envs=${mode#*:}
    env ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0 $envs timeout 600 python bench.py --workload $w --steps 200 --warmup 24 --no-cpu-baseline 2>gpurun_out/r4o/err_${name}_$w.txt | grep '^{' | tail -1 > gpurun_out/r4o/bench_sharded_world1_overlap_${name}_$w.json
    echo "$w overlap $name: $(python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(d["ms_per_step"], d["value"], d["config"].get("overlap","")[:20])' gpurun_out/r4o/bench_sharded_world1_overlap_${name}_$w.json 2>&1 | tail -1)"
  done
done

t() { python bench.py --workload triplet "$@" --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'; }
ESR_STL_PLAN_STREAM=side python -m pytest tests/test_gpu_stl_loop.py tests/test_gpu_triplet_step.py -x -q 2>&1 | tail -2
for b in 128 1024 8192 32768 65536 262144; do
  s=400; [ $b -ge 32768 ] && s=100; [ $b -ge 262144 ] && s=64
  echo "B=$b main: $(t --batch $b --steps $s --warmup 16)   side: $(ESR_STL_PLAN_STREAM=side t --batch $b --steps $s --warmup 16)"
done
echo "zipf 8192 main: $(t --ids zipf --steps 400 --warmup 16)  side: $(ESR_STL_PLAN_STREAM=side t --ids zipf --steps 400 --warmup 16)"

python -m pytest tests/test_gpu_stl_loop.py -x -q 2>&1 | tail -3
for m in "" "ESR_STL_LOOP=presorted"; do for i in 1 2; do
echo "== triplet ${m:-train_steps}: $(env $m python bench.py --workload triplet --steps 800 --warmup 32 --no-cpu-baseline --no-secondary --no-steady --no-kernel-timing 2>/dev/null | grep '^{' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done; done

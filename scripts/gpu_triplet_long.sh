python -m pytest tests/test_gpu_kernels.py -x -q -k "segment_sort_batched" 2>&1 | tail -2
python -m pytest tests/test_gpu_stl_loop.py tests/test_gpu_triplet_step.py -x -q 2>&1 | tail -2
t() { python bench.py --workload triplet "$@" --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["config"]["launch"][:60])'; }
for i in 1 2; do echo "triplet 262144 grouped: $(t --batch 262144 --steps 30 --warmup 5)"; done
echo "triplet 262144 in-line: $(ESR_STL_SORT_BATCH_MAX_IDS=32768 t --batch 262144 --steps 30 --warmup 5)"
echo "triplet 65536 grouped: $(t --batch 65536 --steps 100 --warmup 10)   in-line: $(ESR_STL_SORT_BATCH_MAX_IDS=32768 t --batch 65536 --steps 100 --warmup 10)"
echo "triplet 8192: $(t --steps 400 --warmup 20)"

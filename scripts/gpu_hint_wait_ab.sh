t() { python bench.py --workload triplet --steps 400 --warmup 20 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'; }
g() { python bench.py --workload glove --batch 2048 --steps 400 --warmup 20 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'; }
for i in 1 2; do
echo "triplet wait: $(t)   nowait: $(ESR_STL_HINT_WAIT=0 t)"
echo "glove2048 wait: $(g)   nowait: $(ESR_GLOVE_HINT_WAIT=0 g)"
done
python -m pytest tests/test_gpu_stl_loop.py tests/test_gpu_glove_step.py tests/test_gpu_triplet_step.py -x -q 2>&1 | tail -2

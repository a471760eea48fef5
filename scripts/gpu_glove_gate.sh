python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edges.py tests/test_gpu_retrieve.py -x -q -k "topk or top_k or argsort" 2>&1 | tail -2
python -m pytest tests/test_gpu_glove_step.py -x -q 2>&1 | tail -2
g() { python bench.py --workload glove "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])'; }
for i in 1 2 3; do echo "glove: $(g)   no hint wait: $(ESR_GLOVE_HINT_WAIT=0 g)"; done
echo "glove zipf: $(g --ids zipf)"
echo "glove B=32768: $(g --batch 32768)"

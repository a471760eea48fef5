# GloVe epoch loop after the grouped long-list sort: sort tests, bench at C3 sizes, the timeline
python -m pytest tests/test_gpu_kernels.py -x -q -k "segment_sort_batched" 2>&1 | tail -3
python -m pytest tests/test_gpu_glove_step.py -x -q 2>&1 | tail -3
g() { python bench.py --workload glove "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])'; }
for i in 1 2; do echo "glove grouped: $(g)"; done
echo "glove side-stream: $(ESR_GLOVE_GROUP_SORT_MAX_IDS=32768 g)"
echo "glove zipf grouped: $(g --ids zipf)"
for b in 4096 8192 16384 32768; do echo "glove B=$b grouped: $(g --batch $b)   side-stream: $(ESR_GLOVE_GROUP_SORT_MAX_IDS=4096 g --batch $b)"; done
bash scripts/gpu_timeline_glove.sh

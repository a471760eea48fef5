g() { python bench.py --workload glove "$@" --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])'; }
for b in 262144 524288; do
echo "B=$b grouped(2^21): $(ESR_GLOVE_GROUP_SORT_MAX_IDS=2097152 g --batch $b --steps 40 --warmup 10)   side-stream: $(ESR_GLOVE_GROUP_SORT_MAX_IDS=262144 g --batch $b --steps 40 --warmup 10)"
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d["roofline"]["per_kernel_rocprof"])[:400])'

# round 4: lazy compaction of the running top-k lists (brute-force retrieval): tests, then the C5 leg both ways
python -m pytest tests/test_gpu_retrieve.py tests/test_gpu_ivf.py -x -q 2>&1 | tail -3
for lz in 1 0; do for prec in auto f32; do
  echo "== ESR_RETRIEVE_LAZY=$lz precision=$prec: $(ESR_RETRIEVE_LAZY=$lz python bench.py --workload retrieve --rows 1048576 --precision $prec --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])')"
done; done

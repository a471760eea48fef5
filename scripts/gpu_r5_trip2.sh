mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_stl_loop.py tests/test_gpu_triplet_step.py tests/test_gpu_config_size_oracle.py -k "not glove" -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error" | tail -5) 
for B in 8192 4096 131072; do
  timeout 600 python bench.py --workload triplet --batch $B --steps 400 --warmup 40 --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('triplet B=$B auto', round(d['ms_per_step'],5), round(d['value']/1e6,2))"
done

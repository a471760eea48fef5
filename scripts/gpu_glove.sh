(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -k "glove" 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -5)
for i in 1 2; do (timeout 300 python bench.py --workload glove --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:10], d['value'], d['ms_per_step'], d['config']['loss'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"); done

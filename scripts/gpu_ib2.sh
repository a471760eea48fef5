(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -k "inbatch or stl or towers" 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -8)
for i in 1 2; do (timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'], d['roofline']['frac'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"); done
(timeout 300 python bench.py --no-cpu-baseline --table-dtype bf16 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bf16 tables', d['value'], d['ms_per_step'], d['config']['loss'])")

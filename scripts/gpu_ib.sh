mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -k "inbatch or stl" -s 2>&1 | grep -E "passed|failed|rel.err|Error|assert" | tail -12)
for i in 1 2; do (timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'], d['roofline']['frac'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"); done

mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_sharded.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -15) > gpurun_out/t_k.log 2>&1
tail -4 gpurun_out/t_k.log
for w in inbatch triplet glove; do (timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:20], d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"); done

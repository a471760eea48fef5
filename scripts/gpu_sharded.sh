(timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2)
for w in inbatch triplet glove; do (ESR_BENCH_SHARDED=1 timeout 300 python bench.py --workload $w --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:10], d['value'], d['ms_per_step'], d['config']['loss'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"); done

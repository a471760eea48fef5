for w in inbatch triplet glove; do (timeout 300 python bench.py --workload $w --ids zipf --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('zipf', d['config']['workload'][:10], d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"); done

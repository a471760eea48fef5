(timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|rror" | tail -8)
bash scripts/gpu_zipf.sh
for w in inbatch triplet glove; do (timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('uniform', d['config']['workload'][:10], d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"); done

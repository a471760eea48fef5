for w in inbatch triplet glove; do (timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:10], d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"); done
(timeout 300 python bench.py --workload triplet --graph --no-cpu-baseline --no-kernel-timing 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('triplet graph', d['value'], d['ms_per_step'])")
python benchmarks/spotify_step.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
for w in inbatch triplet glove; do (ESR_BENCH_SHARDED=1 timeout 300 python bench.py --workload $w --steps 100 --warmup 10 --no-kernel-timing 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sharded', d['config']['workload'][:10], d['value'], d['ms_per_step'])"); done

(timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -3)
for d in 1 0; do for w in inbatch triplet glove; do (ESR_RCCL_DIRECT=$d ESR_BENCH_SHARDED=1 timeout 300 python bench.py --workload $w --steps 100 --warmup 10 --no-kernel-timing 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('direct=$d', d['config']['workload'][:10], d['value'], d['ms_per_step'], d['config']['loss'])"); done; done

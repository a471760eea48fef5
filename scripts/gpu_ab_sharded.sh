# A/B of two library builds on the row-sharded leg (world 1): usage bash scripts/gpu_ab_sharded.sh [workload] [rounds]
w=${1:-inbatch}; n=${2:-3}
for r in $(seq 1 $n); do
  for v in old new; do
    cp scripts/ab/$v.so esrecsys_amd/libesr_hip.so
    (ESR_BENCH_SHARDED=1 timeout 300 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w', d['value'], d['ms_per_step'])")
  done
done

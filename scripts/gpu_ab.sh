# A/B of two builds of libesr_hip.so on the same box, alternating: scripts/ab/old.so vs scripts/ab/new.so
# usage: bash scripts/gpu_ab.sh [workload] [rounds]
w=${1:-inbatch}; n=${2:-3}
for r in $(seq 1 $n); do
  for v in old new; do
    cp scripts/ab/$v.so esrecsys_amd/libesr_hip.so
    (timeout 300 python bench.py --workload $w --no-cpu-baseline --no-kernel-timing 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])")
  done
done

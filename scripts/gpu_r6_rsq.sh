for r in 1 2; do
for lib in esrecsys_amd/libesr_hip.so esrecsys_amd/libesr_hip_rsq.so; do
for leg in "--workload triplet --batch 262144" "--workload triplet" "--workload glove" "--workload glove --batch 2048"; do
  ESR_HIP_LIB=$PWD/$lib timeout 600 python bench.py $leg --steps 200 --warmup 16 --no-secondary --no-cpu-baseline --no-steady --no-kernel-timing 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib $leg', round(d['ms_per_step'],5), round(d['value']/1e6,1))"
done
done
done

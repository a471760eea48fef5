mkdir -p gpurun_out
for r in 1 2; do
for leg in "--workload triplet --batch 262144" "--workload triplet --batch 262144 --table-dtype bf16" "--workload glove" "--workload glove --table-dtype bf16"; do
  timeout 600 python bench.py $leg --steps 100 --warmup 16 --no-secondary --no-cpu-baseline --no-steady 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$leg', round(d['ms_per_step'],5), round(d['value']/1e6,1), r.get('frac'), r.get('dominant_kernel'))"
done
done

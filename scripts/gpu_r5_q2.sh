mkdir -p gpurun_out
for q in 32 64; do
  ESR_IB2H_Q=$q timeout 600 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>gpurun_out/q2.err | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('Q rows per wave $q:', round(d['ms_per_step'],5), round(d['value']/1e6,2), d['roofline'].get('per_kernel_us_in_run'))" | tee -a gpurun_out/q2_ab.log
done

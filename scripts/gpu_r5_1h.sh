mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_kernels.py -k "bf16_tables_on_one or gather_folded or one_plane or hard_inputs or fused_towers" tests/test_gpu_stl_loop.py -k "one_call or bf16" -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "worst|AssertionError|Error|passed|failed|FAILED" | cut -c1-600) > gpurun_out/t_1h.log 2>&1
cat gpurun_out/t_1h.log | tail -30
for mode in f16 bf16x3; do
  ESR_INBATCH_BF16_TABLES=$mode timeout 600 python bench.py --table-dtype bf16 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>gpurun_out/1h.err | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 tables path=$mode', round(d['ms_per_step'],5), round(d['value']/1e6,2), d['roofline'].get('per_kernel_us_in_run'), d['roofline'].get('frac'))" | tee -a gpurun_out/inbatch_bf16_ab.log
  tail -2 gpurun_out/1h.err | cut -c1-300
done

mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stl_loop.py tests/test_gpu_config_size_oracle.py -k "inbatch or one_call or bf16 or one_plane or hard_inputs or fused_towers or gather_folded" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/t_mu.log 2>&1
cat gpurun_out/t_mu.log
for mu in 1 0; do
ESR_IB2H_MERGE_UPDATE=$mu timeout 600 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>gpurun_out/mu.err | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp32 tables merge_update=$mu', round(d['ms_per_step'],5), round(d['value']/1e6,2), d['config'].get('loss'), d['roofline'].get('per_kernel_us_in_run'))" | tee -a gpurun_out/mu_bench.log
ESR_IB2H_MERGE_UPDATE=$mu ESR_INBATCH_BF16_TABLES=f16 timeout 600 python bench.py --table-dtype bf16 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>gpurun_out/mu.err | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 tables merge_update=$mu', round(d['ms_per_step'],5), round(d['value']/1e6,2), d['config'].get('loss'), d['roofline'].get('per_kernel_us_in_run'))" | tee -a gpurun_out/mu_bench.log
done

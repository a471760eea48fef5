# round 6: the rebuilt pass C (inbatch2h_pct_kernel): in-batch tests, then the headline leg with per-kernel times
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stl_loop.py tests/test_gpu_api.py -m gpu -q -x -k "inbatch or stl or in_batch" -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40) > gpurun_out/r6_pct_tests.log 2>&1
tail -15 gpurun_out/r6_pct_tests.log
timeout 600 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>gpurun_out/r6_pct_bench.err | grep '^{"metric"' > gpurun_out/r6_pct_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_pct_bench.json').read())
print(d["ms_per_step"], d["value"], d["roofline"].get("per_kernel_us_in_run"))
PY

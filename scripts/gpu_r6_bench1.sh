mkdir -p gpurun_out
timeout 600 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>gpurun_out/r6_bench.err | grep '^{"metric"' > gpurun_out/r6_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench.json').read())
print(d["ms_per_step"], d["value"], d["roofline"].get("per_kernel_us_in_run"))
PY

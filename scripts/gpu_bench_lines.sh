# the driver's command; every JSON line's leg name and size, then the final line
mkdir -p gpurun_out/r4a
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a/bench_driver_cmd.out 2> gpurun_out/r4a/bench_driver_cmd.err
echo "rc=$?"
grep -v amdgpu.ids gpurun_out/r4a/bench_driver_cmd.err | tail -5
python - <<'P'
import json
for line in open("gpurun_out/r4a/bench_driver_cmd.out"):
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(len(line), d.get("leg", "FINAL"), d.get("error", ""))
print(line[:6000])
P

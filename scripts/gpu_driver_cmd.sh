# the driver's exact command, whole line kept
mkdir -p gpurun_out/driver
t0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/driver/line.json
echo "wall $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/driver/line.json'))
print('headline', d['value'], d['ms_per_step'], 'steady', d.get('steady_state'))
for k, v in d.get('secondary', {}).items():
    print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), ((v.get('roofline') or {}).get('step') or {}).get('frac'), v.get('error'))
PY
python3 scripts/startup_probe.py inbatch 2>&1 | grep -v amdgpu.ids > gpurun_out/driver/probe_inbatch.jsonl
python3 - <<'PY'
import json
for l in open('gpurun_out/driver/probe_inbatch.jsonl'):
    d = json.loads(l); print(d['warmup'], d['steps'], d['prewarm'], round(d['ms_per_step'],4), d['gpu_stamp_ms'][:10])
PY

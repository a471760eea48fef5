# the driver's exact command (twice) + the start-up probe
mkdir -p gpurun_out/startup
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/startup/driver_cmd_$i.json; done
python3 bench.py --no-secondary --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/startup/default_200.json
python3 scripts/startup_probe.py inbatch 2>&1 | grep -v amdgpu.ids > gpurun_out/startup/probe_inbatch.jsonl
for f in gpurun_out/startup/driver_cmd_1.json gpurun_out/startup/driver_cmd_2.json gpurun_out/startup/default_200.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'])"; done
cut -c1-1500 gpurun_out/startup/probe_inbatch.jsonl

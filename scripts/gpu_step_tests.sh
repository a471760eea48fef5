# GPU tests of the one-pass steps + a quick bench of the three HBM legs
mkdir -p gpurun_out/steps
timeout 1500 python -m pytest tests/test_gpu_glove_step.py tests/test_gpu_triplet_step.py tests/test_gpu_stl_loop.py -x -q -m gpu 2>&1 | tail -15
for w in triplet glove; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/steps/bench_$w.json; python3 -c "
import json; d=json.load(open('gpurun_out/steps/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['step'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"; done
timeout 300 python bench.py --workload glove --batch 2048 --steps 400 --warmup 20 --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('glove2048', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --workload triplet --batch 262144 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('triplet262144', d['value'], d['ms_per_step'], d['roofline']['step'])"
timeout 300 python bench.py --workload triplet --steps 400 --warmup 20 --ids zipf --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('triplet zipf', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --workload glove --ids zipf --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('glove zipf', d['value'], d['ms_per_step'])"
ESR_GLOVE_PRESORT=0 timeout 300 python bench.py --workload glove --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('glove in-line sort', d['value'], d['ms_per_step'])"
ESR_STL_PLAN_STREAM=side timeout 300 python bench.py --workload triplet --steps 400 --warmup 20 --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('triplet plan on side stream', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --workload triplet --steps 400 --warmup 20 --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('triplet 400 steps', d['value'], d['ms_per_step'])"
ESR_STL_PLAN_STREAM=side timeout 600 python -m pytest tests/test_gpu_stl_loop.py -x -q -m gpu 2>&1 | tail -2

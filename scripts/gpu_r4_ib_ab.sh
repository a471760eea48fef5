# round 4: the fused in-batch head against round 3's launches: tests, then op time and per-kernel durations
python -m pytest tests/test_gpu_kernels.py -x -q -k "inbatch" 2>&1 | tail -4
for v in "ESR_IB2H_FUSED=1" "ESR_IB2H_FUSED=0"; do echo "== $v"; env $v python scripts/ib_ktime.py 8192 300 2>&1 | grep -v amdgpu.ids; done
for f in 1 0; do echo "== bench ESR_IB2H_FUSED=$f"; for i in 1 2; do ESR_IB2H_FUSED=$f python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'; done; done

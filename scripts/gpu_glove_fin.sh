# round 4: finalize inside the GloVe update kernel (short lists): tests, then the B = 2048 leg both ways
python -m pytest tests/test_gpu_glove_step.py tests/test_gpu_api.py -x -q 2>&1 | tail -4
for i in 1 2 3 4 5; do python -m pytest tests/test_gpu_api.py -x -q -k graphed 2>&1 | tail -1; done
for f in 1 0; do echo "== ESR_GLOVE_FIN_FUSED=$f"; for i in 1 2; do ESR_GLOVE_FIN_FUSED=$f python bench.py --workload glove --batch 2048 --steps 800 --warmup 32 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"].get("per_kernel_us_in_run"))'; done; done

python -m pytest tests/test_gpu_kernels.py -x -q -k "segment_sort_batched" 2>&1 | tail -2
python -m pytest tests/test_gpu_glove_step.py -x -q 2>&1 | tail -2
g() { python bench.py --workload glove "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])'; }
for i in 1 2 3; do echo "glove grouped: $(g)"; done
echo "glove zipf grouped: $(g --ids zipf)"
export TMPDIR=/tmp
rm -rf /tmp/tlg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlg -o t -- python bench.py --workload glove --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary --no-steady > /tmp/tlg.log 2>&1
python3 scripts/trace_gaps.py /tmp/tlg radix_tile_batched 2 8 | cut -c1-110

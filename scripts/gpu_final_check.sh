# what the driver runs at round end: GPU tests, smoke, the bench command
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 2>&1 | grep -v "^$" | tail -40 | cut -c1-300) > gpurun_out/t_all.log 2>&1
grep -E "passed|failed" gpurun_out/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench_final.err | grep '^{' > gpurun_out/bench_final_lines.jsonl); tail -1 gpurun_out/bench_final_lines.jsonl | cut -c1-1500; tail -2 gpurun_out/bench_final.err | cut -c1-300

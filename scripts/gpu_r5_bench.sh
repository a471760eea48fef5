# round 5: the driver's command; every JSON line kept
mkdir -p gpurun_out
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench_r5.err | grep "^{" > gpurun_out/bench_r5_lines.jsonl)
tail -c 6000 gpurun_out/bench_r5_lines.jsonl
tail -5 gpurun_out/bench_r5.err

mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -30) > gpurun_out/t_all.log 2>&1
grep -E "passed|failed" gpurun_out/t_all.log
(timeout 300 python bench.py --workload retrieve --steps 10 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/bench_retrieve.json
cut -c1-1500 gpurun_out/bench_retrieve.json
(ESR_BENCH_SHARDED=1 timeout 300 python bench.py --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/bench_sharded_inbatch.json
cut -c1-1500 gpurun_out/bench_sharded_inbatch.json
(timeout 300 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/bench_inbatch.json
cut -c1-700 gpurun_out/bench_inbatch.json

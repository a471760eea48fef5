mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -m gpu -q -s -k "inbatch or stl or graphed" -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40) > gpurun_out/t_inbatch.log 2>&1
for p in f32 bf16x3; do
  (timeout 300 python bench.py --precision $p --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2) > gpurun_out/bench_inbatch_$p.log 2>&1
done
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats_inbatch3 -o inbatch3 -- python bench.py --precision bf16x3 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/prof_stats_inbatch3.log 2>&1
find gpurun_out/prof -name "*.db" -delete; find gpurun_out/prof -name "*kernel_trace.csv" -delete
tail -5 gpurun_out/t_inbatch.log
